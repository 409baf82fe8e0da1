"""Mirror of fl_tissue_model_tools.models for the branching path (reference models.py:597-684).

`UNetXceptionPatchSegmentor` keeps the reference constructor and `predict(x, auto_resample=True)`
contract; the UNet of models.py:85-171 and the smooth tiled prediction run in HIP kernels behind
the C-ABI (tmat_create / tmat_predict_smooth).  Weights come from a "TMATW001" container (Keras
tensor layouts; tools/convert_keras_h5.py turns checkpoint_N.h5 into one where h5py exists).
"""
from __future__ import annotations

import json
import os
from pathlib import Path
from typing import Optional, Tuple

import numpy as np

from . import _lib, synth


class _DeviceModel:
    """Stands where the Keras Model stands in the reference (`segmentor.model`): `.predict(batch)`
    is keras Model.predict on (N, P, P) float32 patches -> (N, P, P, 1)."""

    def __init__(self, handle: _lib.Handle, patch_size: int):
        self.handle = handle
        self.patch_size = patch_size

    def predict(self, batch, verbose=0):
        batch = np.ascontiguousarray(batch, np.float32)
        if batch.ndim == 4 and batch.shape[-1] == 1:
            batch = batch[..., 0]
        if batch.ndim != 3 or batch.shape[1:] != (self.patch_size, self.patch_size):
            raise ValueError(f"expected (N, {self.patch_size}, {self.patch_size}) patches, got {batch.shape}")
        return self.handle.unet_predict(batch)[..., None]


def load_weight_blob(checkpoint_file) -> bytes:
    """bytes / dict of arrays / path to a TMATW001 file.  A Keras .h5 path is accepted when a
    converted '<name>.tmatw' sits next to it."""
    if isinstance(checkpoint_file, (bytes, bytearray)):
        return bytes(checkpoint_file)
    if isinstance(checkpoint_file, dict):
        return synth.pack_weights(checkpoint_file)
    p = Path(checkpoint_file)
    if p.suffix in (".h5", ".hdf5"):
        q = p.with_suffix(".tmatw")
        if q.is_file():
            p = q
        elif os.environ.get("TMAT_SYNTHETIC_WEIGHTS") == "1":
            print(f"[tmat_amd] {p.name} not found/convertible: using synthetic weights (TMAT_SYNTHETIC_WEIGHTS=1)", flush=True)
            return synth.pack_weights(synth.synth_weights(0))
        else:
            raise FileNotFoundError(
                f"{p}: Keras HDF5 checkpoints must be converted once with tools/convert_keras_h5.py "
                f"(expected {q}); set TMAT_SYNTHETIC_WEIGHTS=1 to run with synthetic weights")
    return p.read_bytes()


class UNetXceptionPatchSegmentor:
    """Binary segmentation inference on images in patches with the UNetXception model
    (reference models.py:597-653)."""

    def __init__(self, patch_size: int, checkpoint_file, filter_counts: Tuple[int], ds_ratio: float = 0.5,
                 norm_mean: Optional[float] = None, norm_std: Optional[float] = None, channels: int = 1,
                 device_id: int = 0, max_patches: int = 0):
        if channels != 1:
            raise ValueError("only single-channel models are supported")
        self.patch_size = patch_size
        self.channels = channels
        self.norm_mean = norm_mean
        self.norm_std = norm_std
        self.ds_ratio = ds_ratio
        blob = load_weight_blob(checkpoint_file)
        w = synth.unpack_weights(blob)
        got = sorted({w["stem.w"].shape[-1]} | {v.shape[-1] for k, v in w.items() if k.endswith("sep2.pw")})
        if sorted(filter_counts) != got:
            raise ValueError(f"filter_counts {sorted(filter_counts)} do not match the checkpoint {got}")
        self.handle = _lib.Handle(blob, device_id, max_patches)
        self.model = _DeviceModel(self.handle, patch_size)

    def predict(self, x: np.ndarray, auto_resample=True) -> np.ndarray:
        from .smooth_tiled_predictions import predict_img_with_smooth_windowing
        x = np.asarray(x).astype(np.float32)
        original_shape = x.shape
        target_shape = tuple(np.round(np.multiply(original_shape[:2], self.ds_ratio)).astype(int))
        do_resampling = original_shape != target_shape and auto_resample
        if do_resampling:
            from PIL import Image
            x = np.array(Image.fromarray(x).resize(target_shape, resample=Image.Resampling.LANCZOS))
        if self.norm_mean is not None and self.norm_std is not None:
            x = ((x - self.norm_mean) / self.norm_std).astype(np.float32)
        pred = predict_img_with_smooth_windowing(x, window_size=self.patch_size, subdivisions=2,
                                                 pred_func=self.model.predict)
        if do_resampling:
            from PIL import Image
            pred = np.array(Image.fromarray(pred).resize(original_shape, resample=Image.Resampling.NEAREST))
        return pred


def get_unet_patch_segmentor_from_cfg(cfg_json: str, **kw) -> UNetXceptionPatchSegmentor:
    """reference models.py:656-684; the checkpoint is looked up in ../checkpoints/ next to the
    config directory (MODEL_TRAINING_DIR/binary_segmentation/{configs,checkpoints})."""
    with open(cfg_json, "r") as fp:
        cfg = json.load(fp)
    ckpt = Path(cfg_json).resolve().parent.parent / "checkpoints" / cfg["checkpoint_file"]
    return UNetXceptionPatchSegmentor(cfg["patch_size"], ckpt, cfg["filter_counts"], ds_ratio=cfg.get("ds_ratio", 1),
                                      norm_mean=cfg.get("norm_mean", None), norm_std=cfg.get("norm_std", None),
                                      channels=cfg.get("channels", 1), **kw)
