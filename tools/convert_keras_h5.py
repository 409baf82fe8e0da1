#!/usr/bin/env python3
"""Convert the reference's Keras HDF5 checkpoint (model_training/binary_segmentation/checkpoints/
checkpoint_N.h5, loaded by models.py:622 `load_weights`) into the TMATW001 container read by
tmat_create.  Needs h5py (not present in the build image; run it wherever the checkpoint lives).

    python tools/convert_keras_h5.py checkpoint_1.h5 checkpoint_1.tmatw

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


if __name__ == "__main__":
    convert(sys.argv[1], sys.argv[2])
