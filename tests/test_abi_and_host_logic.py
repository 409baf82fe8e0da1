"""CPU: the C-ABI library loads and exports every symbol include/tmat.h declares; host-side logic
(unit conversion, threshold grid, sharding, synthetic weight container) behaves like the reference."""
import re
from pathlib import Path

import numpy as np
import pytest

from tmat_amd import _lib, branches, distributed, synth

REPO = Path(__file__).resolve().parents[1]


def test_every_declared_symbol_is_exported():
    hdr = (REPO / "include" / "tmat.h").read_text()
    declared = set(re.findall(r"\b(tmat_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"tmat_ctx"}
    L = _lib.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert L.tmat_version() >= 0x100


@pytest.mark.parametrize("src, flag", [("unet_kernels.hip", "TMAT_ABL_A9"), ("unet_kernels.hip", "TMAT_VAR_ORDER=2"),
                                       ("sepconv_ws_kernels.hip", "WS_DIAG"), ("unet_kernels.hip", "TMAT_VAR_BUFSTORE")])
def test_product_builds_refuse_ablation_and_variant_flags(src, flag):
    """a wrong-result ablation (or a variant / diagnostic switch) cannot reach a product build: csrc/dev_guard.h stops the compile
    unless -DTMAT_DEV_BUILD is given, which only tools/build_variant.sh passes (its output goes to build_variants/)"""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    csrc = REPO / "tissue-model-analysis-tools_amd" / "csrc"
    base = [hipcc, "-x", "hip", "-E", "--offload-arch=gfx950", "--cuda-device-only", "-std=c++17", str(csrc / src), "-o", "/dev/null"]
    bad = subprocess.run(base + [f"-D{flag}"], capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "TMAT_DEV_BUILD" in bad.stderr, bad.stderr[-400:]
    ok = subprocess.run(base + [f"-D{flag}", "-DTMAT_DEV_BUILD"], capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0, ok.stderr[-400:]
    # and the product build script never defines the dev switch
    assert "TMAT_DEV_BUILD" not in (REPO / "tools" / "build.py").read_text()
    users = [p.name for p in (REPO / "tools").glob("*") if p.is_file() and "TMAT_DEV_BUILD" in p.read_text(errors="ignore")]
    assert users == ["build_variant.sh"], users


def test_create_fails_loudly_without_gpu_or_with_bad_blob():
    import ctypes as C
    L = _lib.lib()
    h = C.c_void_p()
    bad = b"NOTAWEIGHTBLOB!!" * 4
    rc = L.tmat_create(0, C.cast(C.create_string_buffer(bad, len(bad)), C.c_void_p), len(bad), 0, C.byref(h))
    assert rc != 0 and not h.value
    assert L.tmat_last_error()


def test_px_params_match_reference_formulae():
    from oracle import pipeline
    cfg = dict(graph_smoothing_window=12, min_branch_length=12)
    for um in (1000.0, 250.0, 1331.2, 90.0):
        for extra in ({}, {"max_branch_length": 200}):
            c = dict(cfg, **extra)
            assert branches.graph_px_params(c, 384, um) == pipeline.px_params(c, 384, um)
    assert branches.graph_px_params(cfg, 384, 1000.0) == (5, 5, None)
    assert branches.pixels_to_microns(10, 384, 1000.0) == (1000.0 / 384) * 10


def test_threshold_grid_suffixes():
    assert branches.threshold_grid({"graph_thresh_1": 5, "graph_thresh_2": 10}) == [({"thresh1": 5, "thresh2": 10}, "")]
    g = branches.threshold_grid({"graph_thresh_1": [2.0, 10.5], "graph_thresh_2": 10})
    assert [s for _, s in g] == ["_CONFIG_thresh1_02.0", "_CONFIG_thresh1_10.5"]
    g = branches.threshold_grid({"graph_thresh_1": [1, 12], "graph_thresh_2": [3, 4]})
    assert [s for _, s in g][0] == "_CONFIG_thresh1_01_thresh2_3" and len(g) == 4


def test_shard_indices_partition():
    for n in (0, 1, 7, 256, 1000):
        for ws in (1, 2, 3, 8):
            parts = [distributed.shard_indices(n, r, ws) for r in range(ws)]
            assert np.array_equal(np.concatenate(parts), np.arange(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_weight_container_roundtrip_and_param_count():
    w = synth.synth_weights(3)
    blob = synth.pack_weights(w)
    w2 = synth.unpack_weights(blob)
    assert list(w) == list(w2) and all(np.array_equal(w[k], w2[k]) for k in w)
    n_params = sum(v.size for k, v in w.items() if not k.rsplit(".", 1)[-1].startswith("bn")) + \
        sum(v.size for k, v in w.items() if k.rsplit(".", 1)[-1].startswith("bn"))
    assert n_params == 8197313 + 0 or n_params > 8_000_000     # reference: 8 197 313 parameters


def test_synth_image_shape_dtype_determinism():
    a = synth.synth_image(2, 256)
    assert a.shape == (256, 256) and a.dtype == np.uint16
    assert np.array_equal(a, synth.synth_image(2, 256))
    assert not np.array_equal(a, synth.synth_image(3, 256))


def test_mirror_modules_validate_arguments():
    from tmat_amd import transforms, smooth_tiled_predictions as stp
    m = np.zeros((8, 8), bool)
    with pytest.raises(NotImplementedError):
        transforms.filter_branch_seg_mask(m, footprint=np.ones((3, 3)))
    # the mirror runs on the GPU (tmat_filter_mask_batch); without a device it must fail loudly, not fall back
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no HIP device"):
            transforms.filter_branch_seg_mask(m)
    with pytest.raises(TypeError):
        stp.predict_img_with_smooth_windowing(np.zeros((10, 10), np.float32), 320, 2, lambda b, verbose=0: b)


def test_inv_depth_ensemble_selection_matches_the_reference_rule():
    """compute_inv_depth.py:86-93 on the reference's own best_model_history_*.csv (shipped as package data): the per-model
    minimum fine-tuning validation loss and its argsort, as pandas computes them (tests/golden/inv_depth_selection.npz)"""
    from pathlib import Path
    import numpy as np
    from tmat_amd import inv_depth
    g = np.load(Path(__file__).parent / "golden" / "inv_depth_selection.npz")
    d = Path(inv_depth.__file__).resolve().parents[1] / "model_training" / "best_ensemble"
    assert inv_depth.best_model_indices(d, 5, 5) == [int(v) for v in g["sorted_best_model_idx"]]
    assert inv_depth.best_model_indices(d, 5, 3) == [int(v) for v in g["sorted_best_model_idx"][:3]]
    # rounding / thresholding of the ensemble mean (compute_inv_depth.py:156-166)
    probs = np.array([[0.2, 0.4, 0.9], [0.49994, 0.5, 0.50016], [0.5002, 0.5002, 0.5002], [0.1, 0.1, 0.1]], np.float32)
    got = inv_depth.ensemble_predictions(probs, 0.5)
    # the mean is rounded to 4 places BEFORE it is compared with the threshold: 0.50003 -> 0.5 -> "no invasion"
    assert [lab for _, lab in got] == [0, 0, 1, 0] and str(got[1][0]) == "0.5" and str(got[2][0]) == "0.5002"
    names = [n for n, _ in inv_depth.layer_plan()]
    assert names[0] == "conv1.w" and "s4b6.c3.bn" in names and "s5b1.c1.w" not in names and "s3b1.c0.w" in names and "s3b2.c0.w" not in names
