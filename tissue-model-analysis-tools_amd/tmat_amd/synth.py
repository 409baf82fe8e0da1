"""Synthetic inputs and synthetic UNet weights for benchmarks and tests.

There is no network and the reference's trained checkpoint (checkpoint_1.h5) and sample images
are absent from /root/reference (.MISSING_LARGE_BLOBS), so bench.py and the tests use

* `synth_image(i, size)`: uint16 Z-projection look-alike: noisy background plus 40 random
  cubic-Bezier "vessels", Gaussian blurred (generator fixed by SURVEY.md 8d);
* `synth_weights(seed)`: random-init UNet-Xception weights of the reference architecture
  (models.py:85-171, unet_patch_segmentor_1.json) in Keras tensor layouts.  The weights are
  "structured-synthetic": channel 0 carries a max-pooled / re-smoothed copy of the input through
  the 20x20 bottleneck so the sigmoid output is a vessel-like probability map (every other
  weight is N(0, 1e-3), so every MAC of the architecture is live).

Weight blob container ("TMATW001"): see `pack_weights` / `unpack_weights`; the C-ABI library
(`tmat_create`) parses the same container.
"""
from __future__ import annotations

import struct
from collections import OrderedDict

import numpy as np

FILTER_COUNTS = (64, 128, 256, 512)
PATCH_SIZE = 320
BN_EPS = 1e-3
MAGIC = b"TMATW001"


# --------------------------------------------------------------------------------------
# work accounting (what the GPU path executes on the matrix cores; bench.py's path-level roofline fraction)
# --------------------------------------------------------------------------------------
def mfma_flops_per_patch(filter_counts=None, patch: int = None) -> float:
    """FLOPs per patch that the HIP path runs on MFMA (csrc/tmat_api.cpp schedule): pointwise and residual 1x1
    convolutions (residuals of the up path hoisted below the upsampling), the 3x3 transposed convolutions, and the
    sub-pixel form (16 instead of 36 tap-pairs... i.e. 4 taps per parity class) for those that read an upsampled tensor.
    The stem, the depthwise taps and the final 64 -> 1 convolution run on the vector ALUs and are not counted."""
    f = list(filter_counts or FILTER_COUNTS)
    P = patch or PATCH_SIZE
    fl = 0.0
    H, cin = P // 2, f[0]
    for cout in f[1:]:                       # down blocks
        fl += 2.0 * H * H * cin * cout + 2.0 * H * H * cout * cout + 2.0 * (H // 2) ** 2 * cin * cout
        H //= 2
        cin = cout
    Hs, up = H, 0
    for cout in f[::-1]:                     # up blocks; Hs = stored resolution of the block input
        taps1 = 16 if up else 9              # sub-pixel: 4 classes x 4 taps over the stored pixels
        Hl = Hs << up
        fl += 2.0 * Hs * Hs * taps1 * cin * cout + 2.0 * Hs * Hs * cin * cout + 2.0 * Hl * Hl * 9 * cout * cout
        Hs, up, cin = Hl, 1, cout
    return fl


# --------------------------------------------------------------------------------------
# images
# --------------------------------------------------------------------------------------
def synth_image(index: int, size: int = 1024, n_vessels: int = 40, scale: float = None) -> np.ndarray:
    """uint16 (size, size) synthetic Z-projection, deterministic in `index`.  Vessel widths and blur
    scale with size/1024 unless `scale` is given (scale=1: a crop-like image at full resolution)."""
    from scipy.ndimage import gaussian_filter

    rs = np.random.RandomState(1234 + index)
    img = 2000.0 + rs.normal(0.0, 500.0, (size, size))
    scale = size / 1024.0 if scale is None else float(scale)
    tt = np.linspace(0.0, 1.0, int(700 * scale) + 50)[:, None]
    b0, b1, b2, b3 = (1 - tt) ** 3, 3 * (1 - tt) ** 2 * tt, 3 * (1 - tt) * tt**2, tt**3
    for _ in range(n_vessels):
        ctrl = rs.uniform(0, size, (4, 2))
        width = rs.uniform(3, 9) * scale
        peak = rs.uniform(20000, 50000)
        pts = b0 * ctrl[0] + b1 * ctrl[1] + b2 * ctrl[2] + b3 * ctrl[3]
        r = max(width / 2.0, 0.75)
        ri = int(np.ceil(r)) + 1
        yy, xx = np.mgrid[-ri : ri + 1, -ri : ri + 1]
        vmask = np.zeros((size, size), dtype=bool)
        for py, px in pts:
            cy, cx = int(round(py)), int(round(px))
            y0, y1 = max(cy - ri, 0), min(cy + ri + 1, size)
            x0, x1 = max(cx - ri, 0), min(cx + ri + 1, size)
            if y0 >= y1 or x0 >= x1:
                continue
            disc = (yy[y0 - cy + ri : y1 - cy + ri, x0 - cx + ri : x1 - cx + ri] + cy - py) ** 2 + (
                xx[y0 - cy + ri : y1 - cy + ri, x0 - cx + ri : x1 - cx + ri] + cx - px
            ) ** 2 <= r * r
            vmask[y0:y1, x0:x1] |= disc
        img += peak * vmask
    img = gaussian_filter(img, 1.5 * scale)
    return np.clip(img, 0, 65535).astype(np.uint16)


def synth_stack(index: int, n_slices: int = 12, height: int = 512, width: int = 512, n_vessels: int = 14) -> np.ndarray:
    """uint16 (Z, height, width) synthetic confocal-like Z stack, deterministic in `index`: curved tubes that each live in a
    few neighbouring slices (gaussian profile along Z), plus sensor noise"""
    from scipy.ndimage import gaussian_filter

    rs = np.random.RandomState(4321 + index)
    scale = width / 1024.0
    vol = np.zeros((n_slices, height, width), np.float64)
    tt = np.linspace(0.0, 1.0, int(900 * scale) + 60)[:, None]
    b0, b1, b2, b3 = (1 - tt) ** 3, 3 * (1 - tt) ** 2 * tt, 3 * (1 - tt) * tt**2, tt**3
    yy, xx = np.mgrid[:height, :width]
    for _ in range(n_vessels):
        ctrl = rs.uniform(0, 1, (4, 2)) * (height, width)
        pts = b0 * ctrl[0] + b1 * ctrl[1] + b2 * ctrl[2] + b3 * ctrl[3]
        plane = np.zeros((height, width))
        iy = np.clip(np.round(pts[:, 0]).astype(int), 0, height - 1)
        ix = np.clip(np.round(pts[:, 1]).astype(int), 0, width - 1)
        plane[iy, ix] = 1.0
        plane = gaussian_filter(plane, rs.uniform(3.0, 8.0) * scale + 0.8)
        plane /= plane.max()
        zc, zw = rs.uniform(0, n_slices - 1), rs.uniform(1.0, 2.5)
        prof = np.exp(-(((np.arange(n_slices) - zc) / zw) ** 2))
        vol += rs.uniform(15000, 45000) * prof[:, None, None] * plane[None]
    vol += 1500.0 + rs.normal(0.0, 400.0, vol.shape)
    return np.clip(vol, 0, 65535).astype(np.uint16)


# --------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------
def layer_plan(filter_counts=FILTER_COUNTS):
    """[(name, shape)] of every tensor of the architecture, Keras layouts."""
    f = sorted(filter_counts)
    plan = [("stem.w", (3, 3, 1, f[0])), ("stem.b", (f[0],)), ("stem.bn", (4, f[0]))]
    cin = f[0]
    for i, fo in enumerate(f[1:]):
        p = f"down{i}"
        plan += [
            (f"{p}.sep1.dw", (3, 3, cin)), (f"{p}.sep1.pw", (cin, fo)), (f"{p}.sep1.b", (fo,)),
            (f"{p}.bn1", (4, fo)),
            (f"{p}.sep2.dw", (3, 3, fo)), (f"{p}.sep2.pw", (fo, fo)), (f"{p}.sep2.b", (fo,)),
            (f"{p}.bn2", (4, fo)),
            (f"{p}.res.w", (cin, fo)), (f"{p}.res.b", (fo,)),
        ]
        cin = fo
    for j, fo in enumerate(reversed(f)):
        p = f"up{j}"
        plan += [
            (f"{p}.ct1.w", (3, 3, fo, cin)), (f"{p}.ct1.b", (fo,)), (f"{p}.bn1", (4, fo)),
            (f"{p}.ct2.w", (3, 3, fo, fo)), (f"{p}.ct2.b", (fo,)), (f"{p}.bn2", (4, fo)),
            (f"{p}.res.w", (cin, fo)), (f"{p}.res.b", (fo,)),
        ]
        cin = fo
    plan += [("final.w", (3, 3, cin)), ("final.b", (1,))]
    return plan


def synth_weights(seed: int = 0, filter_counts=FILTER_COUNTS, noise: float = 1e-3, sat_thresh: float = 0.25,
                  sat_gain: float = 6.0, out_gain: float = 12.0, out_bias: float = -8.0) -> "OrderedDict[str, np.ndarray]":
    """Random-init weights of the reference architecture with a hand-built channel-0 signal path:
    stem 3x3 box -> soft binarisation min(max(g (a - t), 0), 1) (two ReLU stages of down block 0) ->
    three max-pools (keeps thin vessels connected down to the 20x20 bottleneck) -> [1 2 1]^2/16
    smoothing in every transposed conv of the up path -> final conv gain/bias -> sigmoid.
    All other weights are N(0, noise): every MAC of the architecture is live.  The decoder of this
    architecture (nearest upsampling + translation-invariant convs) cannot be hand-wired to emit
    detail finer than the 16-px bottleneck grid, so the synthetic masks are coarser than a trained
    model's; see DESIGN.md "Synthetic workload"."""
    rs = np.random.RandomState(seed)
    w = OrderedDict()
    for name, shape in layer_plan(filter_counts):
        if name.rsplit(".", 1)[-1].startswith("bn"):
            bn = np.zeros(shape, np.float32)
            bn[0] = 1.0 + rs.normal(0, 0.01, shape[1])   # gamma
            bn[1] = rs.normal(0, noise, shape[1])        # beta
            bn[2] = rs.normal(0, noise, shape[1])        # moving mean
            bn[3] = 1.0 + rs.uniform(0, 0.02, shape[1])  # moving variance
            bn[:, 0] = (1.0, 0.0, 0.0, 1.0 - BN_EPS)     # channel 0: identity
            w[name] = bn.astype(np.float32)
        else:
            w[name] = rs.normal(0, noise, shape).astype(np.float32)
    smooth = np.outer([1, 2, 1], [1, 2, 1]).astype(np.float32) / 16.0
    w["stem.w"][:, :, 0, 0] = 1.0 / 9.0
    w["stem.b"][0] = 0.0
    for i in range(len(filter_counts) - 1):
        p = f"down{i}"
        for s in ("sep1", "sep2"):
            w[f"{p}.{s}.dw"][:, :, 0] = 0.0
            w[f"{p}.{s}.dw"][1, 1, 0] = 1.0
            w[f"{p}.{s}.pw"][0, 0] = 1.0
            w[f"{p}.{s}.b"][0] = 0.0
        w[f"{p}.res.w"][0, 0] = 0.0
        w[f"{p}.res.b"][0] = 0.0
    # soft binarisation in down block 0: u = relu(1 - g (a - t)); v = 1 - u
    w["down0.sep1.pw"][0, 0] = -sat_gain
    w["down0.sep1.b"][0] = 1.0 + sat_gain * sat_thresh
    w["down0.sep2.pw"][0, 0] = -1.0
    w["down0.sep2.b"][0] = 1.0
    for j in range(len(filter_counts)):
        p = f"up{j}"
        for s in ("ct1", "ct2"):
            w[f"{p}.{s}.w"][:, :, 0, 0] = smooth
            w[f"{p}.{s}.b"][0] = 0.0
        w[f"{p}.res.w"][0, 0] = 0.0
        w[f"{p}.res.b"][0] = 0.0
    w["final.w"][:, :, 0] = out_gain * smooth
    w["final.b"][0] = out_bias
    return w


def pack_weights(w: "OrderedDict[str, np.ndarray]", patch_size=PATCH_SIZE) -> bytes:
    """Container: magic(8) | u32 n | u32 patch_size | n * { name[48] | u32 ndim | u32 dims[4] |
    u64 offset | u64 count } | f32 payload (each tensor 16-byte aligned).  Little endian."""
    ent = 48 + 4 + 16 + 8 + 8
    head = 8 + 4 + 4 + ent * len(w)
    off = (head + 15) // 16 * 16
    table, payload = b"", bytearray()
    base = off
    for name, a in w.items():
        a = np.ascontiguousarray(a, np.float32)
        dims = list(a.shape) + [1] * (4 - a.ndim)
        table += struct.pack("<48sI4IQQ", name.encode(), a.ndim, *dims, base + len(payload), a.size)
        payload += a.tobytes()
        payload += b"\0" * ((-len(payload)) % 16)
    blob = MAGIC + struct.pack("<II", len(w), patch_size) + table
    blob += b"\0" * (off - len(blob))
    return bytes(blob) + bytes(payload)


def unpack_weights(blob: bytes) -> "OrderedDict[str, np.ndarray]":
    assert blob[:8] == MAGIC, "bad weight blob magic"
    n, _ps = struct.unpack_from("<II", blob, 8)
    w = OrderedDict()
    pos = 16
    for _ in range(n):
        name, ndim, d0, d1, d2, d3, off, cnt = struct.unpack_from("<48sI4IQQ", blob, pos)
        pos += 84
        shape = (d0, d1, d2, d3)[:ndim]
        w[name.rstrip(b"\0").decode()] = np.frombuffer(blob, np.float32, cnt, off).reshape(shape).copy()
    return w
