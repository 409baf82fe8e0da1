#!/usr/bin/env python3
"""Convert the reference's Keras HDF5 checkpoint (model_training/binary_segmentation/checkpoints/
checkpoint_N.h5, loaded by models.py:622 `load_weights`) into the TMATW001 container read by
tmat_create.  Needs h5py (not present in the build image; run it wherever the checkpoint lives).

    python tools/convert_keras_h5.py checkpoint_1.h5 checkpoint_1.tmatw
    python tools/convert_keras_h5.py --resnet best_finetune_weights_0.h5 best_finetune_weights_0.tmatw     (invasion-depth classifier)

Layer order follows build_UNetXception (models.py:110-166); tensors keep their Keras layouts
(SURVEY.md A1).  The file is opened read-only with h5py: nothing in it is executed.
"""
import sys
from collections import OrderedDict
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tissue-model-analysis-tools_amd"))


def _layers(f):
    g = f["model_weights"] if "model_weights" in f else f
    names = [n.decode() if isinstance(n, bytes) else n for n in g.attrs["layer_names"]]
    out = []
    for n in names:
        wn = [w.decode() if isinstance(w, bytes) else w for w in g[n].attrs.get("weight_names", [])]
        if wn:
            out.append((n, [np.array(g[n][w]) for w in wn]))
    return out


def convert(src, dst):
    import h5py
    from tmat_amd import synth
    with h5py.File(src, "r") as f:
        layers = _layers(f)
    kinds = lambda pre: [(n, w) for n, w in layers if n.startswith(pre)]
    conv, sep, ct, bn = kinds("conv2d"), kinds("separable_conv2d"), kinds("conv2d_transpose"), kinds("batch_normalization")
    conv = [c for c in conv if not c[0].startswith("conv2d_transpose")]
    w = OrderedDict()
    ci, si, ti, bi = iter(conv), iter(sep), iter(ct), iter(bn)
    nb = lambda: np.stack(next(bi)[1]).astype(np.float32)                 # gamma, beta, mean, var
    k, b = next(ci)[1]
    w["stem.w"], w["stem.b"], w["stem.bn"] = k, b, nb()
    n_down = len(sep) // 2
    for i in range(n_down):
        p = f"down{i}"
        for s, bname in (("sep1", "bn1"), ("sep2", "bn2")):
            dw, pw, bias = next(si)[1]
            w[f"{p}.{s}.dw"], w[f"{p}.{s}.pw"], w[f"{p}.{s}.b"] = dw[..., 0], pw[0, 0], bias
            w[f"{p}.{bname}"] = nb()
        k, b = next(ci)[1]
        w[f"{p}.res.w"], w[f"{p}.res.b"] = k[0, 0], b
    for j in range(len(ct) // 2):
        p = f"up{j}"
        for s, bname in (("ct1", "bn1"), ("ct2", "bn2")):
            k, b = next(ti)[1]
            w[f"{p}.{s}.w"], w[f"{p}.{s}.b"], w[f"{p}.{bname}"] = k, b, nb()
        k, b = next(ci)[1]
        w[f"{p}.res.w"], w[f"{p}.res.b"] = k[0, 0], b
    k, b = next(ci)[1]
    w["final.w"], w["final.b"] = k[..., 0], b
    plan = dict(synth.layer_plan(sorted({w["stem.w"].shape[-1]} | {v.shape[-1] for n, v in w.items() if n.endswith("sep2.pw")})))
    for n, v in w.items():
        assert tuple(v.shape) == tuple(plan[n]), (n, v.shape, plan[n])
    Path(dst).write_bytes(synth.pack_weights(OrderedDict((n, np.asarray(v, np.float32)) for n, v in w.items())))
    print(f"wrote {dst}: {sum(v.size for v in w.values())} parameters")


def convert_resnet(src, dst):
    """best_finetune_weights_N.h5 of the invasion-depth ensemble (compute_inv_depth.py:114-118: build_ResNet50_TL(...).load_weights):
    the nested `base_model` holds keras.applications' layer names (conv1_conv, conv1_bn, conv{S}_block{B}_{k}_conv / _bn with
    k = 0 for the projection shortcut), the head is `dense`.  Every dataset is addressed by its "<layer>/<variable>:0" tail."""
    import re
    import h5py
    from tmat_amd import inv_depth
    tensors = {}
    with h5py.File(src, "r") as f:
        def visit(name, obj):
            if isinstance(obj, h5py.Dataset):
                parts = name.split("/")
                tensors["/".join(parts[-2:])] = np.array(obj)
        f.visititems(visit)
    w = OrderedDict()

    def take(keras_layer, mine):
        bn = keras_layer.replace("_conv", "_bn")
        w[mine + ".w"] = tensors[f"{keras_layer}/kernel:0"]
        w[mine + ".b"] = tensors[f"{keras_layer}/bias:0"]
        w[mine + ".bn"] = np.stack([tensors[f"{bn}/{v}:0"] for v in ("gamma", "beta", "moving_mean", "moving_variance")]).astype(np.float32)
    take("conv1_conv", "conv1")
    blocks = sorted({(int(m.group(1)), int(m.group(2))) for m in (re.match(r"conv(\d)_block(\d)_1_conv/kernel:0", k) for k in tensors) if m})
    for stage, blk in blocks:
        for k in (1, 2, 3) + ((0,) if blk == 1 else ()):
            take(f"conv{stage}_block{blk}_{k}_conv", f"s{stage}b{blk}.c{k}")
    dense = [k for k in tensors if re.match(r"dense(_\d+)?/kernel:0", k)]
    assert len(dense) == 1, dense
    w["fc.w"] = tensors[dense[0]].reshape(-1)
    w["fc.b"] = tensors[dense[0].replace("kernel", "bias")].reshape(-1)
    stage, blk = blocks[-1]
    plan = inv_depth.layer_plan(f"conv{stage}_block{blk}_out")
    ordered = OrderedDict((n, np.asarray(w[n], np.float32)) for n, _ in plan)
    for n, shape in plan:
        assert tuple(ordered[n].shape) == tuple(shape), (n, ordered[n].shape, shape)
    Path(dst).write_bytes(inv_depth.pack_resnet(ordered))
    print(f"wrote {dst}: {sum(v.size for v in ordered.values())} parameters")


if __name__ == "__main__":
    if sys.argv[1] == "--resnet":
        convert_resnet(sys.argv[2], sys.argv[3])
    else:
        convert(sys.argv[1], sys.argv[2])
