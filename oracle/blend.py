"""oracle/blend.py -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

numpy restatement of the smooth tiled prediction driver
(reference fl_tissue_model_tools/smooth_tiled_predictions.py):
  spline window      :26-65      pad / unpad     :68-92
  D4 do / undo       :95-133     tile + predict  :136-192
  overlap-add        :195-217    driver          :220-267
Arithmetic order is the reference's: f32 patch * f64 window, f64 overlap-add in row-major tile
order, / subdivisions**2, D4 undo, sequential sum of the 8 orientations, / 8.
Pinned bit-exactly against the imported reference by tests/golden/blend_*.npz
(tools/make_goldens.py).
"""
from __future__ import annotations

import numpy as np


def triang(M: int) -> np.ndarray:
    """scipy.signal.windows.triang(M) (symmetric) restated."""
    n = np.arange(1, (M + 1) // 2 + 1)
    if M % 2 == 0:
        w = (2 * n - 1.0) / M
        return np.r_[w, w[::-1]]
    w = 2 * n / (M + 1.0)
    return np.r_[w, w[-2::-1]]


def spline_window(window_size: int, power: int = 2) -> np.ndarray:
    """smooth_tiled_predictions.py:26-41."""
    intersection = int(window_size / 4)
    tri = triang(window_size)
    outer = (abs(2 * tri) ** power) / 2
    outer[intersection:-intersection] = 0
    inner = 1 - (abs(2 * (tri - 1)) ** power) / 2
    inner[:intersection] = 0
    inner[-intersection:] = 0
    wind = inner + outer
    return wind / np.average(wind)


def window_2d(window_size: int) -> np.ndarray:
    """:47-65 without the trailing channel axis: w[i] * w[j]."""
    w = spline_window(window_size)
    return w[:, None] * w[None, :]


def d4_do(im: np.ndarray):
    """:95-113"""
    m = im[:, ::-1]
    return [np.rot90(im, k) for k in range(4)] + [np.rot90(m, k) for k in range(4)]


def d4_undo_mean(res):
    """:116-133: rotate/mirror back, then np.mean over the 8 (sequential f64 sum, / 8)."""
    origs = [np.rot90(res[k], (4 - k) % 4) for k in range(4)]
    origs += [np.rot90(res[4 + k], (4 - k) % 4)[:, ::-1] for k in range(4)]
    acc = np.array(origs[0], dtype=np.float64)
    for o in origs[1:]:
        acc = acc + o
    return acc / 8.0


def predict_img_with_smooth_windowing(input_img, window_size, subdivisions, pred_func, batch=16):
    """:220-267 for a 2-D input image and a single-channel pred_func
    (pred_func(batch (N,ws,ws) f32, verbose=0) -> (N,ws,ws,1) f32)."""
    aug = int(round(window_size * (1 - 1.0 / subdivisions)))
    step = int(window_size / subdivisions)
    pad = np.pad(input_img, ((aug, aug), (aug, aug)), mode="constant", constant_values=input_img.min())
    win = window_2d(window_size)
    res = []
    for p in d4_do(pad):
        H, W = p.shape
        ii = list(range(0, H - window_size + 1, step))
        jj = list(range(0, W - window_size + 1, step))
        tiles = np.array([p[i:i + window_size, j:j + window_size] for i in ii for j in jj])
        preds = np.concatenate([pred_func(tiles[k:k + batch], verbose=0) for k in range(0, len(tiles), batch)])
        preds = preds[..., 0] * win                       # f32 * f64 -> f64
        y = np.zeros((H, W))
        t = 0
        for i in ii:
            for j in jj:
                y[i:i + window_size, j:j + window_size] = y[i:i + window_size, j:j + window_size] + preds[t]
                t += 1
        res.append(y / (subdivisions ** 2))
    out = d4_undo_mean(res)
    out = out[aug:-aug, aug:-aug]
    return out[: input_img.shape[0], : input_img.shape[1]]
