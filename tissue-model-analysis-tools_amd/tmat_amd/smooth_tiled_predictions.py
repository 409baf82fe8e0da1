"""Mirror of fl_tissue_model_tools.smooth_tiled_predictions (reference :220-267).

The reference tiles, predicts, windows and blends in numpy around `pred_func`.  Here the whole
driver is one device pipeline (tile gather -> UNet -> f64 window blend, csrc/blend_kernels.hip), so
`pred_func` must be the `.predict` of a device model (models.UNetXceptionPatchSegmentor.model).
"""
from __future__ import annotations

import numpy as np

INFERENCE_BATCH_SIZE = 16   # kept for API compatibility; the device path batches whole images


def predict_img_with_smooth_windowing(input_img, window_size, subdivisions, pred_func):
    model = getattr(pred_func, "__self__", None)
    handle = getattr(model, "handle", None)
    if handle is None:
        raise TypeError("tmat_amd.predict_img_with_smooth_windowing needs pred_func = <segmentor>.model.predict "
                        "(a device model); arbitrary Python predictors are not supported by the HIP path")
    if subdivisions != 2:
        raise ValueError("only subdivisions=2 (the reference's value, models.py:643) is supported")
    if window_size != model.patch_size:
        raise ValueError("window_size must equal the model's patch size")
    img = np.asarray(input_img)
    if img.ndim == 3 and img.shape[-1] == 1:
        img = img[..., 0]
    if img.ndim != 2:
        raise ValueError("expected a 2-D single-channel image")
    return handle.predict_smooth(img.astype(np.float32))
