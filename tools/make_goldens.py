#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference's own Python modules from
/root/reference (read-only, this container only; the reference never travels to the GPU box).

Only inputs and expected outputs are stored -- no reference source text.
Run:  python tools/make_goldens.py [blend] [dmt] [morse] [filter]
The `filter` group needs scikit-image and therefore /opt/conda/bin/python3.9 (skimage 0.18.3).
Shims for absent third-party modules (numba.njit = identity decorator, cv2 colour stub used only
by plotting code) are created in a temp dir, exactly as SURVEY.md section 8c describes.
"""
import hashlib
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
GOLD = REPO / "tests" / "golden"
REF = Path("/root/reference")


def _shims():
    d = Path(tempfile.mkdtemp(prefix="tmat_shims_"))
    (d / "numba.py").write_text(
        "def njit(*a, **k):\n"
        "    if a and callable(a[0]):\n        return a[0]\n"
        "    return lambda f: f\n")
    (d / "cv2.py").write_text(
        "COLOR_HSV2BGR = 54\nINTER_LANCZOS4 = 4\n"
        "def cvtColor(a, code):\n    return a\n")
    sys.path.insert(0, str(d))
    sys.path.insert(0, str(REF))


def pred_toy(batch, verbose=0):
    b = np.asarray(batch, np.float32)
    out = (0.6 * b + 0.3 * b[:, ::-1, :] * b[:, :, ::-1] + 0.05).astype(np.float32)
    return out[..., None]


def gen_blend():
    from fl_tissue_model_tools import smooth_tiled_predictions as stp

    out = {}
    out["window320"] = stp._spline_window(320)
    out["window64"] = stp._spline_window(64)
    rs = np.random.RandomState(7)
    for name, shape, ws in (("toy_a", (100, 90), 64), ("toy_b", (64, 64), 32), ("toy_c", (75, 130), 32)):
        img = rs.uniform(0, 1, shape).astype(np.float32)
        stp.cached_2d_windows.clear()
        res = stp.predict_img_with_smooth_windowing(img, ws, 2, pred_toy)
        out[name + "_in"] = img
        out[name + "_out"] = np.asarray(res)
        out[name + "_ws"] = np.int64(ws)
    img = np.random.RandomState(11).uniform(0, 1, (640, 640)).astype(np.float32)
    stp.cached_2d_windows.clear()
    res = stp.predict_img_with_smooth_windowing(img, 320, 2, lambda b, verbose=0: np.asarray(b)[..., None])
    out["ident640_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(res).tobytes()).digest(), np.uint8)
    out["ident640_maxerr"] = np.float64(np.abs(res - img).max())
    np.savez_compressed(GOLD / "blend.npz", **out)
    print("blend.npz", {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    GOLD.mkdir(parents=True, exist_ok=True)
    _shims()
    groups = sys.argv[1:] or ["blend"]
    for g in groups:
        globals()["gen_" + g]()
