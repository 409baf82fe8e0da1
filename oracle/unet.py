"""oracle/unet.py -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU oracle of the UNet-Xception inference graph (reference models.py:110-166) on (N, 320, 320)
float32 patch batches, i.e. what `keras_model.predict` does for
smooth_tiled_predictions.py:179/182.

* `forward_exact(w, x)`: fixed-order fmaf-chain arithmetic in C (oracle/unet_exact.c), the
  bit-exact parity target of the HIP kernels.
* `forward_torch(w, x)`: independent as-written restatement with PyTorch-CPU ops
  (conv_transpose2d, upsample-then-1x1 residual, un-folded batch-norm); used to validate the
  algebraic rewrites of the exact path (ConvT flip, residual hoisting, BN folding) and as the fast
  all-cores CPU baseline timed by bench.py.

TensorFlow 2.14.1 (setup.py:73) is absent, so both are "parity unpinned" against TF itself;
layer semantics follow SURVEY.md Appendix A1.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
BUILD = HERE / "_build"
BN_EPS = 1e-3
_lib = None


def build(force: bool = False) -> Path:
    """Compile oracle/*.c into oracle/_build/liboracle.so (gcc only)."""
    BUILD.mkdir(exist_ok=True)
    so = BUILD / "liboracle.so"
    srcs = sorted(HERE.glob("*.c"))
    if not force and so.exists() and all(so.stat().st_mtime >= s.stat().st_mtime for s in srcs):
        return so
    cmd = ["gcc", "-O3", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-fno-math-errno",
           "-fno-trapping-math", "-o", str(so)] + [str(s) for s in srcs] + ["-lm"]
    subprocess.run(cmd, check=True)
    return so


def usable_cores() -> int:
    """cores this process may really use (affinity mask, capped at 16 = one GPU's share of the box)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 8
    return max(1, min(n, int(os.environ.get("TMAT_ORACLE_THREADS", "16"))))


def lib():
    global _lib
    if _lib is None:
        os.environ.setdefault("OMP_NUM_THREADS", str(usable_cores()))
        _lib = ctypes.CDLL(str(build()))
        _lib.orc_sigmoid.restype = ctypes.c_float
        _lib.orc_sigmoid.argtypes = [ctypes.c_float]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


# ----------------------------------------------------------------------------------------
# weight preparation (the same folding the C-ABI library performs in tmat_create)
# ----------------------------------------------------------------------------------------
def fold_bn(bn, bias):
    """scale = f32(gamma / sqrt(var + eps)); shift = f32(beta + (bias - mean) * scale_f64)."""
    g, b, m, v = (bn[i].astype(np.float64) for i in range(4))
    sd = g / np.sqrt(v + BN_EPS)
    return sd.astype(np.float32), (b + (bias.astype(np.float64) - m) * sd).astype(np.float32)


def convt_as_conv(k):
    """Keras Conv2DTranspose kernel (kh, kw, out, in), stride 1, SAME  ->  conv taps
    Wc[a][b][in][out] = K[2-a][2-b][out][in]   (SURVEY.md A1)."""
    return _f32(k[::-1, ::-1].transpose(0, 1, 3, 2))


def subpixel_weights(Wc):
    """3x3 conv taps (3, 3, Cin, Cout) -> (4, 4, Cin, Cout): the taps of a 3x3 convolution over a 2x nearest-upsampled
    tensor that land on the same stored pixel, summed per output parity class (py, px).  Full-resolution offset d of an
    output pixel of parity p reads stored offset (p + d) >> 1, i.e. slot a = ((p + d) >> 1) - (p - 1).  f32 adds from
    +0.0 in (ky, kx) order -- the same order as csrc/tmat_api.cpp:subpixel_weights."""
    Wc = _f32(Wc)
    out = np.zeros((2, 2, 2, 2) + Wc.shape[2:], np.float32)
    for py in range(2):
        for px in range(2):
            for ky in range(3):
                for kx in range(3):
                    a = ((py + ky - 1) >> 1) - (py - 1)
                    b = ((px + kx - 1) >> 1) - (px - 1)
                    out[py, px, a, b] = out[py, px, a, b] + Wc[ky, kx]
    return out.reshape((4, 4) + Wc.shape[2:])


# ----------------------------------------------------------------------------------------
# exact path
# ----------------------------------------------------------------------------------------
def _conv_subpixel(S, Wc3, relu_in, scale, shift, relu_out):
    L = lib()
    N, h, w, cin = S.shape
    cout = Wc3.shape[-1]
    out = np.empty((N, 2 * h, 2 * w, cout), np.float32)
    Ws = _f32(subpixel_weights(Wc3))
    rc = L.orc_conv_subpixel(_p(S), N, h, w, cin, relu_in, _p(Ws), cout, _p(scale), _p(shift), relu_out, _p(out))
    assert rc == 0, rc
    return out


def _conv(S, Wc, ksize, stride, up, relu_in, scale, shift, resid, rs, relu_out):
    L = lib()
    N, h, w, cin = S.shape
    cout = Wc.shape[-1]
    H, W = (h << up) // stride, (w << up) // stride
    out = np.empty((N, H, W, cout), np.float32)
    Wc = _f32(Wc.reshape(ksize * ksize, cin, cout))
    rc = L.orc_conv(_p(S), N, h, w, cin, up, relu_in, _p(Wc), ksize, stride, cout, _p(scale), _p(shift),
                    _p(resid), rs, relu_out, _p(out))
    assert rc == 0, rc
    return out


def _dw(S, Wd, relu_in):
    L = lib()
    N, H, W, C = S.shape
    out = np.empty_like(S)
    rc = L.orc_dwconv(_p(S), N, H, W, C, relu_in, _p(_f32(Wd.reshape(9, C))), _p(out))
    assert rc == 0
    return out


def forward_exact(w, x, taps=None, subpixel=True):
    """x: (N, P, P) float32 -> (N, P, P) float32 sigmoid output.  `taps` (dict) collects
    intermediate tensors by name when given.  `subpixel=False` evaluates the first transposed convolution of the
    up blocks as the as-written 9-tap chain over the upsampled tensor instead of the 4-tap sub-pixel form (the two
    differ by f32 rounding of the pre-summed taps only; tests bound the difference)."""
    L = lib()
    x = _f32(x)
    N, P, _ = x.shape
    f0 = w["stem.w"].shape[-1]
    sc, sh = fold_bn(w["stem.bn"], w["stem.b"])
    a = np.empty((N, P // 2, P // 2, f0), np.float32)
    L.orc_stem(_p(x), N, P, P, _p(_f32(w["stem.w"].reshape(9, f0))), f0, _p(sc), _p(sh), _p(a))
    if taps is not None:
        taps["stem"] = a
    prev = a
    i = 0
    while f"down{i}.sep1.dw" in w:
        p = f"down{i}"
        d1 = _dw(prev, w[f"{p}.sep1.dw"], 1)
        sc, sh = fold_bn(w[f"{p}.bn1"], w[f"{p}.sep1.b"])
        p1 = _conv(d1, w[f"{p}.sep1.pw"], 1, 1, 0, 0, sc, sh, None, 0, 1)
        d2 = _dw(p1, w[f"{p}.sep2.dw"], 0)
        sc, sh = fold_bn(w[f"{p}.bn2"], w[f"{p}.sep2.b"])
        p2 = _conv(d2, w[f"{p}.sep2.pw"], 1, 1, 0, 0, sc, sh, None, 0, 0)
        r = _conv(prev, w[f"{p}.res.w"], 1, 2, 0, 0, None, _f32(w[f"{p}.res.b"]), None, 0, 0)
        Nn, H, W, C = p2.shape
        out = np.empty((Nn, H // 2, W // 2, C), np.float32)
        L.orc_maxpool_add(_p(p2), Nn, H, W, C, _p(r), _p(out))
        prev = out
        if taps is not None:
            taps[p] = out
        i += 1
    S, up = prev, 0
    j = 0
    while f"up{j}.ct1.w" in w:
        p = f"up{j}"
        sc, sh = fold_bn(w[f"{p}.bn1"], w[f"{p}.ct1.b"])
        if up and subpixel:
            t1 = _conv_subpixel(S, convt_as_conv(w[f"{p}.ct1.w"]), 1, sc, sh, 1)
        else:
            t1 = _conv(S, convt_as_conv(w[f"{p}.ct1.w"]), 3, 1, up, 1, sc, sh, None, 0, 1)
        rr = _conv(S, w[f"{p}.res.w"], 1, 1, 0, 0, None, _f32(w[f"{p}.res.b"]), None, 0, 0)
        sc, sh = fold_bn(w[f"{p}.bn2"], w[f"{p}.ct2.b"])
        S = _conv(t1, convt_as_conv(w[f"{p}.ct2.w"]), 3, 1, 0, 0, sc, sh, rr, up, 0)
        if taps is not None:
            taps[p] = S
        up = 1
        j += 1
    Nn, h, ww, C = S.shape
    out = np.empty((Nn, 2 * h, 2 * ww), np.float32)
    wsub = _f32(subpixel_weights(w["final.w"].reshape(3, 3, C, 1)).reshape(4, 4, C))
    L.orc_final(_p(S), Nn, h, ww, C, _p(wsub), ctypes.c_float(float(w["final.b"][0])), _p(out))
    return out


def predict_exact(w, chunk=16):
    """pred_func(batch, verbose=0) -> (N, P, P, 1), the Keras Model.predict contract used by
    smooth_tiled_predictions.py:179."""
    def pred(batch, verbose=0):
        batch = np.asarray(batch, np.float32)
        outs = [forward_exact(w, batch[i:i + chunk]) for i in range(0, len(batch), chunk)]
        return np.concatenate(outs)[..., None]
    return pred


# ----------------------------------------------------------------------------------------
# independent torch restatement (as written in models.py:110-166)
# ----------------------------------------------------------------------------------------
def forward_torch(w, x, dtype=None):
    import torch
    import torch.nn.functional as F

    dt = dtype or torch.float32
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dt)

    def bn(t, p):
        g, b, m, v = (T(p[i]) for i in range(4))
        return F.batch_norm(t, m, v, g, b, training=False, eps=BN_EPS)

    def same_pad_s2(t, k, value=0.0):  # TF SAME, stride 2: pad_total = max(k - 2 + (in % 2... ), even dims
        H, W = t.shape[-2:]
        ph = max((-(-H // 2) - 1) * 2 + k - H, 0)
        pw = max((-(-W // 2) - 1) * 2 + k - W, 0)
        return F.pad(t, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2), value=value)

    with torch.no_grad():
        t = T(x)[:, None]                                               # NCHW
        t = F.conv2d(same_pad_s2(t, 3), T(w["stem.w"]).permute(3, 2, 0, 1), T(w["stem.b"]), stride=2)
        t = F.relu(bn(t, w["stem.bn"]))
        prev = t
        i = 0
        while f"down{i}.sep1.dw" in w:
            p = f"down{i}"
            if i != 0:
                t = F.relu(t)
            for s, b in (("sep1", "bn1"), ("sep2", "bn2")):
                dw = T(w[f"{p}.{s}.dw"]).permute(2, 0, 1)[:, None]      # (C,1,3,3)
                t = F.conv2d(t, dw, None, padding=1, groups=t.shape[1])
                t = F.conv2d(t, T(w[f"{p}.{s}.pw"]).t()[:, :, None, None], T(w[f"{p}.{s}.b"]))
                t = bn(t, w[f"{p}.{b}"])
                if s == "sep1":
                    t = F.relu(t)
            t = F.max_pool2d(same_pad_s2(t, 3, float("-inf")), 3, 2)
            r = F.conv2d(prev, T(w[f"{p}.res.w"]).t()[:, :, None, None], T(w[f"{p}.res.b"]), stride=2)
            t = t + r
            prev = t
            i += 1
        j = 0
        while f"up{j}.ct1.w" in w:
            p = f"up{j}"
            for s, b in (("ct1", "bn1"), ("ct2", "bn2")):
                t = F.relu(t)
                k = T(w[f"{p}.{s}.w"]).permute(3, 2, 0, 1)               # (in,out,kh,kw)
                t = F.conv_transpose2d(t, k, T(w[f"{p}.{s}.b"]), padding=1)
                t = bn(t, w[f"{p}.{b}"])
            t = F.interpolate(t, scale_factor=2, mode="nearest")
            r = F.interpolate(prev, scale_factor=2, mode="nearest")
            r = F.conv2d(r, T(w[f"{p}.res.w"]).t()[:, :, None, None], T(w[f"{p}.res.b"]))
            t = t + r
            prev = t
            j += 1
        k = T(w["final.w"]).permute(2, 0, 1)[None]                       # (1,C,3,3)
        t = torch.sigmoid(F.conv2d(t, k, T(w["final.b"]), padding=1))
    return t[:, 0].numpy()


def predict_torch(w, chunk=16):
    def pred(batch, verbose=0):
        batch = np.asarray(batch, np.float32)
        outs = [forward_torch(w, batch[i:i + chunk]) for i in range(0, len(batch), chunk)]
        return np.concatenate(outs).astype(np.float32)[..., None]
    return pred
