"""oracle/pipeline.py -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

analyze_img's 2-D branch (reference scripts/compute_branches.py:307-361, 391-426, 455-464) restated
end to end from the oracle pieces: Lanczos4 + rescale (oracle/morph.py), smooth tiled UNet
prediction (oracle/blend.py + oracle/unet.py), mask filtering / centre-line weighting / resize
(oracle/morph.py), DMT graph (oracle/dmt.c), MorseGraph statistics (oracle/morse.py).

`unet="exact"` is the bit-exact parity target of the HIP path; `unet="torch"` is the fast all-core
CPU baseline timed by bench.py (same stages, PyTorch-CPU convolutions).
"""
from __future__ import annotations

import numpy as np

from . import blend, dmt, morph, morse, unet, wellmask

DOWNSAMPLE_WIDTH = 384


def px_params(config: dict, field_width: int, image_width_microns: float):
    """compute_branches.py:401-415: microns -> pixels of the 384-wide field."""
    to_px = lambda um: (field_width / image_width_microns) * um
    min_px = round(to_px(config.get("min_branch_length", 12)))
    mx = config.get("max_branch_length")
    max_px = None if mx is None else round(max(1, to_px(mx)))
    sw_px = round(max(1, to_px(config.get("graph_smoothing_window", 12))))
    return sw_px, min_px, max_px


def segment(img_u16: np.ndarray, weights, ds_ratio=0.625, unet_kind="exact", patch=320, input_bits=16):
    tgt = morph.resized_shape(img_u16.shape, ds_ratio)       # cv2's dsize is (width, height): see morph.target_shape
    small = morph.lanczos4_resize(img_u16, tgt, input_bits)
    x = morph.rescale_intensity(small, (0, 1)).astype(np.float32)
    pf = unet.predict_exact(weights) if unet_kind == "exact" else unet.predict_torch(weights)
    return blend.predict_img_with_smooth_windowing(x, patch, 2, pf)


def analyze_image(img_u16: np.ndarray, weights, config: dict, image_width_microns: float, ds_ratio=0.625,
                  unet_kind="exact", return_intermediates=False, input_bits=16):
    """-> (count, total_px, avg_px) for one image and one (thresh1, thresh2) pair."""
    pred = segment(img_u16, weights, ds_ratio, unet_kind, input_bits=input_bits)
    out_shape = morph.dsamp_shape(img_u16.shape, DOWNSAMPLE_WIDTH)
    field, seg, skel = morph.postprocess(pred, out_shape)
    f255 = morph.rescale_intensity(field, (0, 255))
    t1 = float(np.atleast_1d(config.get("graph_thresh_1", 5))[0])
    t2 = float(np.atleast_1d(config.get("graph_thresh_2", 10))[0])
    sw_px, min_px, max_px = px_params(config, field.shape[1], image_width_microns)
    V, E = dmt.compute_dmt_graph(f255.astype(np.float32), t1, t2)
    bars, n, tot, avg = morse.morse_stats(V, E, field.shape, sw_px, min_px, max_px,
                                          bool(config.get("remove_isolated_branches", False)), None)
    if return_intermediates:
        return (n, tot, avg), dict(pred=pred, field=field, seg=seg, skel=skel, V=V, E=E, bars=bars)
    return n, tot, avg


def analyze_image_well(img_u16: np.ndarray, weights, config: dict, image_width_microns: float, ds_ratio=0.625, seed=0,
                       input_bits=16, norm=None):
    """analyze_img's 2-D branch with use_well_mask=True (compute_branches.py:309-361, 391-457): the well mask is made from the
    downsampled, rescaled image, multiplies the network input and the segmentation mask, and its shrunken form prunes the
    graph.  `norm` = (norm_mean, norm_std) of the model config (models.py:636-637).  -> ((count, total_px, avg_px), masks)"""
    tgt = morph.resized_shape(img_u16.shape, ds_ratio)
    x = morph.rescale_intensity(morph.lanczos4_resize(img_u16, tgt, input_bits), (0, 1)).astype(np.float32)
    well, shrunk = wellmask.make_well_mask(x, seed=seed)
    xin = x * well
    if norm is not None:
        xin = ((xin - norm[0]) / norm[1]).astype(np.float32)
    pred = blend.predict_img_with_smooth_windowing(xin, 320, 2, unet.predict_exact(weights))
    seg = morph.filter_branch_seg_mask((pred > 0.5) & well)
    weighted, skel, _ = morph.centerline_weight(pred, seg.astype(float))
    out_shape = morph.dsamp_shape(img_u16.shape, DOWNSAMPLE_WIDTH)
    field = morph.resize_aa(weighted, out_shape).astype(np.float32)
    pruning = wellmask.resize_nearest(np.logical_not(shrunk), out_shape).astype(bool)
    f255 = morph.rescale_intensity(field, (0, 255))
    t1 = float(np.atleast_1d(config.get("graph_thresh_1", 5))[0])
    t2 = float(np.atleast_1d(config.get("graph_thresh_2", 10))[0])
    sw_px, min_px, max_px = px_params(config, field.shape[1], image_width_microns)
    V, E = dmt.compute_dmt_graph(f255.astype(np.float32), t1, t2)
    _, n, tot, avg = morse.morse_stats(V, E, field.shape, sw_px, min_px, max_px, bool(config.get("remove_isolated_branches", False)), pruning)
    return (n, tot, avg), dict(well=well, shrunk=shrunk, pruning=pruning, pred=pred, field=field)
