"""Host side of the invasion-depth tool (reference scripts/compute_inv_depth.py:50-172, models.py:33-82 build_ResNet50_TL,
data_prep.py:17-61) on the HIP library (csrc/resnet_kernels.hip; no CPU fallback).

The reference builds n_pred_models Keras ResNet50 classifiers, loads `best_finetune_weights_{i}.h5` of the models with the
lowest fine-tuning validation loss, and averages their per-slice probabilities.  Here a classifier is a "TMATW001" weight
container (pack_resnet) loaded with tmat_resnet_load; the data preparation and every model's forward pass for a whole Z stack
are one call (tmat_inv_depth_predict).  The ensemble selection and the rounding / thresholding of the mean are host logic,
as in the reference.
"""
from __future__ import annotations

import csv
import ctypes as C
from collections import OrderedDict
from pathlib import Path

import numpy as np

from . import synth
from ._lib import Handle, check, lib, ptr

STAGES = {2: (64, 3), 3: (128, 4), 4: (256, 6), 5: (512, 3)}          # keras.applications.resnet50: stage -> (width, blocks)
LAST_LAYER_DEFAULT = "conv4_block6_out"                                  # model_training/invasion_depth_best_hp.json


def layer_plan(last_layer: str = LAST_LAYER_DEFAULT):
    """tensor names and Keras shapes of ResNet50 up to `conv{stage}_block{blocks}_out` + the dense head"""
    import re
    m = re.fullmatch(r"conv(\d)_block(\d)_out", last_layer)
    if not m:
        raise ValueError(f"unsupported last_resnet_layer {last_layer}")
    last_stage, last_block = int(m.group(1)), int(m.group(2))
    plan = [("conv1.w", (7, 7, 3, 64)), ("conv1.b", (64,)), ("conv1.bn", (4, 64))]
    cin = 64
    for stage in range(2, last_stage + 1):
        f, nblk = STAGES[stage]
        for blk in range(1, (last_block if stage == last_stage else nblk) + 1):
            p = f"s{stage}b{blk}"
            convs = [("c1", 1, cin, f), ("c2", 3, f, f), ("c3", 1, f, 4 * f)] + ([("c0", 1, cin, 4 * f)] if blk == 1 else [])
            for name, k, ci, co in convs:
                plan += [(f"{p}.{name}.w", (k, k, ci, co)), (f"{p}.{name}.b", (co,)), (f"{p}.{name}.bn", (4, co))]
            cin = 4 * f
    plan += [("fc.w", (cin,)), ("fc.b", (1,))]
    return plan


def flops_per_slice(weights, size: int = 256) -> float:
    """convolution FLOPs (2 per multiply-add) of one forward of the classifier on a size x size slice: conv1 at size / 2, the
    max-pool halves it again, stages 3.. halve it in their first block (stride in c1 and in the shortcut c0)"""
    total, res = 0.0, size // 2
    kh, kw, ci, co = weights["conv1.w"].shape
    total += 2.0 * res * res * kh * kw * ci * co
    res //= 2
    stage_seen = set()
    for name, w in weights.items():
        if not name.endswith(".w") or name in ("conv1.w", "fc.w"):
            continue
        blk, conv = name.split(".")[0], name.split(".")[1]
        stage, b = int(blk[1]), int(blk[3:])
        if stage > 2 and b == 1 and stage not in stage_seen:
            stage_seen.add(stage)
            res //= 2                            # this stage's first block strides (its c1 and c0 already produce the halved map)
        kh, kw, ci, co = w.shape
        total += 2.0 * res * res * kh * kw * ci * co
    return total


def synth_resnet_weights(seed: int = 0, last_layer: str = LAST_LAYER_DEFAULT) -> "OrderedDict[str, np.ndarray]":
    """random-init weights of the architecture (there is no network for ImageNet / fine-tuned checkpoints): He-normal
    convolutions, BatchNormalization statistics near identity with a damped last BN per block so 13 residual blocks keep the
    activations O(1), a dense head that spreads the probabilities over (0, 1)"""
    rs = np.random.RandomState(1000 + seed)
    w = OrderedDict()
    for name, shape in layer_plan(last_layer):
        kind = name.rsplit(".", 1)[1]
        if kind == "w" and name != "fc.w":
            fan_in = shape[0] * shape[1] * shape[2]
            w[name] = rs.normal(0, np.sqrt(2.0 / fan_in), shape).astype(np.float32)
        elif kind == "b":
            w[name] = rs.normal(0, 0.05, shape).astype(np.float32)
        elif kind == "bn":
            C_ = shape[1]
            g = rs.uniform(0.8, 1.2, C_) * (0.3 if ".c3." in name else 1.0)
            w[name] = np.stack([g, rs.normal(0, 0.1, C_), rs.normal(0, 0.1, C_), rs.uniform(0.8, 1.2, C_)]).astype(np.float32)
        else:
            w[name] = rs.normal(0, 0.08, shape).astype(np.float32)
    if seed >= 0:
        w["conv1.w"] = (w["conv1.w"] / 40.0).astype(np.float32)          # inputs are 0..255 minus the caffe means
    return w


def pack_resnet(w) -> bytes:
    return synth.pack_weights(w, patch_size=0)


class InvDepthEnsemble:
    """the classifiers of compute_inv_depth.py:96-121 on one handle"""

    def __init__(self, handle: Handle, weight_sets, size: int = 256):
        self.handle, self.size, self.ids = handle, int(size), []
        for w in weight_sets:
            blob = w if isinstance(w, (bytes, bytearray)) else pack_resnet(w)
            buf = (C.c_char * len(blob)).from_buffer_copy(blob)
            mid = C.c_int()
            check(lib().tmat_resnet_load(handle.raw, C.cast(buf, C.c_void_p), len(blob), C.byref(mid)), "tmat_resnet_load")
            self.ids.append(mid.value)

    def predict(self, x: np.ndarray, model: int = 0) -> np.ndarray:
        """model.predict(x).squeeze() for prepared inputs (n, size, size, 3) float32"""
        x = np.ascontiguousarray(x, np.float32)
        prob = np.empty(len(x), np.float32)
        check(lib().tmat_resnet_predict(self.handle.raw, self.ids[model], ptr(x), len(x), x.shape[1], ptr(prob)), "tmat_resnet_predict")
        return prob

    def predict_stack(self, stack: np.ndarray, return_input=False):
        """(Z, H, W) uint8 / uint16 -> probabilities (Z, n_models) float32 (the transposed yhatp_m of compute_inv_depth.py:154)"""
        a = np.asarray(stack)
        if a.ndim != 3 or a.dtype not in (np.uint8, np.uint16):
            raise ValueError("predict_stack: expected a (Z, H, W) uint8 or uint16 stack")
        from .preprocessing import _source_depth
        src_dtype = a.dtype
        a = np.ascontiguousarray(a, np.uint16)
        ids = np.ascontiguousarray(self.ids, np.int32)
        probs = np.empty((a.shape[0], len(ids)), np.float32)
        x = np.empty((a.shape[0], self.size, self.size, 3), np.float32) if return_input else None
        with _source_depth(self.handle, src_dtype):          # uint8 slices: cv2's fixed-point bilinear arithmetic (data_prep.py:36)
            check(lib().tmat_inv_depth_predict(self.handle.raw, ptr(ids), len(ids), ptr(a), a.shape[0], a.shape[1], a.shape[2], self.size, ptr(probs),
                                               ptr(x) if return_input else None), "tmat_inv_depth_predict")
        return (probs, x) if return_input else probs


    def predict_stacks(self, stacks, max_slices: int = 128):
        """several stacks -> list of (Z_i, n_models) probabilities.  Every step of the tool is per slice (data_prep.py:17-61,
        compute_inv_depth.py:150-154), so stacks of one shape and dtype are concatenated along Z and go through the classifiers
        together, up to `max_slices` per call: larger launches, same per-slice results as predict_stack on each stack."""
        out = [None] * len(stacks)
        i = 0
        while i < len(stacks):
            a0 = np.asarray(stacks[i])
            if a0.ndim != 3 or a0.dtype not in (np.uint8, np.uint16):
                raise ValueError("predict_stacks: expected (Z, H, W) uint8 or uint16 stacks")
            j, nsl = i + 1, a0.shape[0]
            while j < len(stacks) and np.asarray(stacks[j]).shape[1:] == a0.shape[1:] and np.asarray(stacks[j]).dtype == a0.dtype \
                    and nsl + np.asarray(stacks[j]).shape[0] <= max_slices:
                nsl += np.asarray(stacks[j]).shape[0]
                j += 1
            if j > i + 1:
                # one call, no concatenation on the host: the library uploads the stacks back to back (tmat_inv_depth_predict_multi)
                from .preprocessing import _source_depth
                group = [np.ascontiguousarray(np.asarray(s), np.uint16) for s in stacks[i:j]]
                ptrs = (C.c_void_p * len(group))(*[g.ctypes.data for g in group])
                zs = np.ascontiguousarray([g.shape[0] for g in group], np.int32)
                ids = np.ascontiguousarray(self.ids, np.int32)
                probs = np.empty((nsl, len(ids)), np.float32)
                with _source_depth(self.handle, a0.dtype):
                    check(lib().tmat_inv_depth_predict_multi(self.handle.raw, ptr(ids), len(ids), C.cast(ptrs, C.c_void_p), ptr(zs), len(group),
                                                             a0.shape[1], a0.shape[2], self.size, ptr(probs)), "tmat_inv_depth_predict_multi")
            else:
                probs = self.predict_stack(a0)
            z0 = 0
            for k in range(i, j):
                zk = np.asarray(stacks[k]).shape[0]
                out[k] = probs[z0:z0 + zk]
                z0 += zk
            i = j
        return out


def best_model_indices(best_ensemble_dir, n_models: int, n_pred_models: int):
    """compute_inv_depth.py:86-93: the n_pred_models models with the lowest fine-tuning validation loss"""
    best = np.zeros(n_models)
    for i in range(n_models):
        with open(Path(best_ensemble_dir) / f"best_model_history_{i}.csv", newline="") as f:
            rows = [r for r in csv.DictReader(f) if r["training_stage"] == "finetune"]
        best[i] = min(float(r["val_loss"]) for r in rows)
    return [int(v) for v in best.argsort()[:n_pred_models]]


def ensemble_predictions(probs: np.ndarray, cls_thresh: float = 0.5):
    """compute_inv_depth.py:156-166: mean over the models (float32), round to 4 places, label = probability > cls_thresh"""
    yhatp = np.mean(np.asarray(probs, np.float32), axis=1, keepdims=True)
    out = []
    for z in range(len(yhatp)):
        p = round(np.atleast_1d(yhatp[z])[0], 4)
        out.append((p, int(p > cls_thresh)))
    return out
