"""oracle/resnet.py -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU restatement of the invasion-depth classifier (reference fl_tissue_model_tools/models.py:33-82 build_ResNet50_TL,
data_prep.py:17-61, scripts/compute_inv_depth.py:150-166).  The network is keras.applications.resnet50.ResNet50 (tensorflow
2.14.1, reference setup.py:73) cut at `conv4_block6_out` (model_training/invasion_depth_best_hp.json) + GlobalAveragePooling2D
+ Dense(1) + sigmoid.  TensorFlow is absent and the reference holds neither weights nor fixtures for this model (the .h5 files
of model_training/best_ensemble are not in the checkout): PARITY UNPINNED against Keras; the graph below follows the published
architecture (ResNet v1: bias in every convolution, BatchNormalization eps 1.001e-5, ZeroPadding2D(3) + 7x7/2 'valid' stem,
ZeroPadding2D(1) + 3x3/2 'valid' max-pool, bottleneck blocks with the stride in the first 1x1 convolution and in the projection
shortcut of a stage's first block).  What the reference's own files do pin is the ensemble selection: the
best_model_history_*.csv files are read exactly as compute_inv_depth.py:86-93 reads them (tests/golden/inv_depth_histories).

Arithmetic: 1x1 / 3x3 convolutions AND the 7x7 stem (as a 1x1 convolution over its im2col tensor, 147 taps padded to K = 192) through
oracle/unet_exact.c:orc_conv (the chain order of the MFMA kernel); pool and head in float32 numpy with multiply and add kept
separate, in the order csrc/resnet_kernels.hip documents.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import unet as ou

BN_EPS = 1.001e-5
MEANS_BGR = (103.939, 116.779, 123.68)
STAGES = {2: (64, 3), 3: (128, 4), 4: (256, 6), 5: (512, 3)}          # stage -> (bottleneck width, blocks)
F32 = np.float32


def fold_bn(bn, bias):
    g, b, m, v = (bn[i].astype(np.float64) for i in range(4))
    sd = g / np.sqrt(v + BN_EPS)
    return sd.astype(np.float32), (b + (bias.astype(np.float64) - m) * sd).astype(np.float32)


def prep_inv_depth_imgs(stack: np.ndarray, size: int = 256) -> np.ndarray:
    """data_prep.prep_inv_depth_imgs: per slice cv2.resize (bilinear, see oracle/cellarea.py), rescale_intensity(0..255) in
    float64, three identical channels, caffe preprocessing (BGR means), float32"""
    from . import cellarea as ca
    out = np.empty((len(stack), size, size, 3), np.float32)
    for z, sl in enumerate(stack):
        small = ca.resize_linear(sl, (size, size)).astype(np.float64)           # uint8 slices: cv2's fixed-point path
        lo, hi = small.min(), small.max()
        g = np.clip(small, lo, hi)
        g = ((g - lo) / (hi - lo)) * 255.0 + 0.0 if lo != hi else np.clip(g, 0, 255)
        for c in range(3):
            out[z, :, :, c] = (g - MEANS_BGR[c]).astype(np.float32)
    return out


STEM_TAPS, STEM_K = 147, 192


def stem_im2col(x):
    """(N, S, S, 3) -> (N, S/2, S/2, 192): the 147 taps of ZeroPadding2D(3) + Conv2D(64, 7, strides 2) per output pixel, k = (ky 7 + kx) 3 + c,
    zeros outside the image and for k >= 147 (csrc/resnet_kernels.hip:resnet_im2col_kernel)"""
    N, S = x.shape[0], x.shape[1]
    So = S // 2
    xp = np.zeros((N, S + 6, S + 6, 3), np.float32)
    xp[:, 3:-3, 3:-3] = x
    col = np.zeros((N, So, So, STEM_K), np.float32)
    for ky in range(7):
        for kx in range(7):
            k = (ky * 7 + kx) * 3
            col[..., k:k + 3] = xp[:, ky:ky + 2 * So:2, kx:kx + 2 * So:2]
    return col


def stem(x, w, scale, shift):
    """ZeroPadding2D(3) + Conv2D(64, 7, strides 2) + BN + ReLU as a 1x1 convolution over the im2col tensor: orc_conv's chain over
    k (the order conv_mfma_kernel accumulates in), fmaf(acc, scale, shift), ReLU"""
    wk = np.zeros((1, 1, STEM_K, 64), np.float32)
    wk[0, 0, :STEM_TAPS] = w.reshape(STEM_TAPS, 64)
    return ou._conv(stem_im2col(np.ascontiguousarray(x, np.float32)), wk, 1, 1, 0, 0, scale, shift, None, 0, 1)


def pool(x):
    """ZeroPadding2D(1) + MaxPooling2D(3, strides 2)"""
    N, S, _, C = x.shape
    So = S // 2
    xp = np.zeros((N, S + 2, S + 2, C), np.float32)
    xp[:, 1:-1, 1:-1] = x
    m = np.full((N, So, So, C), -np.inf, np.float32)
    for ky in range(3):
        for kx in range(3):
            m = np.maximum(m, xp[:, ky:ky + 2 * So:2, kx:kx + 2 * So:2])
    return m


def head(feat, w, b):
    N, h, _, C = feat.shape
    f = feat.reshape(N, h * h, C)
    s = np.zeros((N, C), np.float32)
    for p in range(h * h):
        s = s + f[:, p]
    s = s * F32(1.0 / (h * h))
    z = np.zeros(N, np.float32)
    for k in range(C):
        z = z + s[:, k] * F32(w[k])
    z = z + F32(b)
    L = ou.lib()
    L.orc_sigmoid.restype = ctypes.c_float
    return np.array([L.orc_sigmoid(ctypes.c_float(float(v))) for v in z], np.float32)


def forward(w: dict, x: np.ndarray) -> np.ndarray:
    """x (N, S, S, 3) float32 -> probabilities (N,) float32"""
    x = np.ascontiguousarray(x, np.float32)
    sc, sh = fold_bn(w["conv1.bn"], w["conv1.b"])
    a = pool(stem(x, w["conv1.w"].astype(np.float32), sc, sh))
    stage = 2
    while f"s{stage}b1.c1.w" in w:
        blk = 1
        while f"s{stage}b{blk}.c1.w" in w:
            p = f"s{stage}b{blk}"
            stride = 2 if (blk == 1 and stage > 2) else 1

            def conv(name, inp, ksize, st, resid=None, relu=1):
                s_, h_ = fold_bn(w[name + ".bn"], w[name + ".b"])
                return ou._conv(np.ascontiguousarray(inp), w[name + ".w"].astype(np.float32), ksize, st, 0, 0, s_, h_, resid, 0, relu)
            shortcut = conv(p + ".c0", a, 1, stride, None, 0) if blk == 1 else a
            t = conv(p + ".c1", a, 1, stride)
            t = conv(p + ".c2", t, 3, 1)
            a = conv(p + ".c3", t, 1, 1, np.ascontiguousarray(shortcut), 1)
            blk += 1
        stage += 1
    return head(a, w["fc.w"].ravel(), float(w["fc.b"].ravel()[0]))


def best_models(histories, n_pred_models):
    """compute_inv_depth.py:86-93: per model the minimum val_loss of its 'finetune' rows, argsort, the first n_pred_models.
    `histories`: list of (val_loss array, training_stage array)"""
    best = np.zeros(len(histories))
    for i, (val_loss, stage) in enumerate(histories):
        best[i] = np.min(np.asarray(val_loss, np.float64)[np.asarray(stage) == "finetune"])
    return [int(v) for v in best.argsort()[:n_pred_models]]


def ensemble(probs: np.ndarray, cls_thresh=0.5):
    """compute_inv_depth.py:156-166: probs (Z, n_models) float32 -> (rounded mean probability, label) per slice"""
    yhatp = np.mean(probs.astype(np.float32), axis=1, keepdims=True)
    out = []
    for z in range(len(yhatp)):
        p = round(np.atleast_1d(yhatp[z])[0], 4)
        out.append((p, int(p > cls_thresh)))
    return out
