"""Batched analyze_img (reference scripts/compute_branches.py:144-489, 2-D branch) over the C-ABI."""
from __future__ import annotations

import ctypes as C
from itertools import product

import numpy as np

from . import _lib

DOWNSAMPLE_WIDTH = 384


def pixels_to_microns(num_pixels: float, im_width_px: int, im_width_microns: float) -> float:
    return (im_width_microns / im_width_px) * num_pixels


def microns_to_pixels(num_microns: float, im_width_px: int, im_width_microns: float) -> float:
    return (im_width_px / im_width_microns) * num_microns


def graph_px_params(config: dict, field_width: int, image_width_microns: float):
    """compute_branches.py:401-415"""
    min_px = round(microns_to_pixels(config.get("min_branch_length", 12), field_width, image_width_microns))
    mx = config.get("max_branch_length")
    max_px = None if mx is None else round(max(1, microns_to_pixels(mx, field_width, image_width_microns)))
    sw_px = round(max(1, microns_to_pixels(config.get("graph_smoothing_window", 12), field_width, image_width_microns)))
    return sw_px, min_px, max_px


def threshold_grid(config: dict):
    """compute_branches.py:366-395: Cartesian grid over graph_thresh_1 x graph_thresh_2 with the file-name suffix."""
    params = {"thresh1": np.atleast_1d(config.get("graph_thresh_1", 5)).tolist(),
              "thresh2": np.atleast_1d(config.get("graph_thresh_2", 10)).tolist()}
    names, vals = zip(*params.items())
    cfgs = [dict(zip(names, comb)) for comb in product(*vals)]
    tuned = [k for k, v in params.items() if len(v) > 1]
    fmts = {}
    for k, v in params.items():
        if all(isinstance(x, (int, float)) for x in v):
            if all(isinstance(x, int) for x in v):
                fmts[k] = f"{{:0{max(len(str(x)) for x in v)}d}}"
            else:
                wl = max(str(float(x)).find(".") for x in v)
                wr = max(len(str(float(x)).split(".")[1]) for x in v)
                fmts[k] = f"{{:0{wl + 1 + wr}.{wr}f}}"
        else:
            fmts[k] = "{}"
    out = []
    for cfg in cfgs:
        s = "".join(f"_{k}_{fmts[k].format(v)}" for k, v in cfg.items() if k in tuned)
        out.append((cfg, f"_CONFIG{s}" if s else ""))
    return out


def analyze_batch(handle: _lib.Handle, imgs: np.ndarray, config: dict, image_width_microns: float, ds_ratio: float = 0.625,
                  thresh=(5.0, 10.0), first_index: int = 0, dev_ptr=None, input_bits: int = 16):
    """imgs (n, H, W) uint16 (host) or a device pointer + shape -> list of (index, count, total_px, avg_px).
    `input_bits` = 8 for images that were uint8 before widening (cv2.resize saturates to the source depth)."""
    if dev_ptr is None:
        imgs = np.ascontiguousarray(imgs, np.uint16)
        n, H, W = imgs.shape
    else:
        n, H, W = imgs       # shape tuple
    sw_px, min_px, max_px = graph_px_params(config, DOWNSAMPLE_WIDTH, image_width_microns)
    rows = (_lib.Row * n)()
    L = _lib.lib()
    _lib.check(L.tmat_set_input_depth(handle.raw, int(input_bits)), "tmat_set_input_depth")
    args = (n, H, W, float(ds_ratio), DOWNSAMPLE_WIDTH, float(thresh[0]), float(thresh[1]), int(sw_px), int(min_px),
            int(max_px or 0), int(bool(config.get("remove_isolated_branches", False))), int(first_index), rows)
    if dev_ptr is None:
        _lib.check(L.tmat_analyze_batch(handle.raw, _lib.ptr(imgs), *args), "tmat_analyze_batch")
    else:
        _lib.check(L.tmat_analyze_batch_dev(handle.raw, C.c_void_p(dev_ptr), *args), "tmat_analyze_batch_dev")
    return [(r.index, r.count, r.total_px, r.avg_px) for r in rows]
