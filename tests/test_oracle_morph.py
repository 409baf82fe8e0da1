"""CPU: oracle/morph.py against golden vectors produced by the reference's transforms.py running on
scikit-image 0.18.3 (tools/make_goldens.py filter)."""
import hashlib

import numpy as np
import pytest

from oracle import morph

G = np.load(__import__("pathlib").Path(__file__).parent / "golden" / "filter.npz")
NAMES = ["d5", "m1", "blobs", "noise", "small", "empty", "full"]


def unpack(name, key):
    shape = tuple(G[name + "_shape"])
    return np.unpackbits(G[name + "_" + key])[: shape[0] * shape[1]].reshape(shape).astype(bool)


@pytest.mark.parametrize("name", NAMES)
def test_median_label_perimeter_skeleton(name):
    m = unpack(name, "mask")
    med = morph.median13(m)
    assert np.array_equal(med, unpack(name, "median"))
    assert np.array_equal(morph.skeletonize_zhang(med), unpack(name, "skel"))
    from scipy import ndimage as ndi
    lab, n = ndi.label(med, np.ones((3, 3)))
    assert n == int(G[name + "_nlabels"])
    areas = np.array([int((lab[sl] == i).sum()) for i, sl in enumerate(ndi.find_objects(lab), 1)], np.int64)
    perims = np.array([morph.perimeter4(lab[sl] == i) for i, sl in enumerate(ndi.find_objects(lab), 1)])
    assert np.array_equal(areas, G[name + "_areas"])
    np.testing.assert_allclose(perims, G[name + "_perims"], rtol=1e-13, atol=0)


@pytest.mark.parametrize("name", NAMES)
def test_filter_branch_seg_mask(name):
    med = unpack(name, "median")
    assert np.array_equal(morph.filter_branch_seg_mask(med, use_median=False), unpack(name, "filtered"))
    assert np.array_equal(morph.filter_branch_seg_mask(med, use_median=False, remove_isolated=False),
                          unpack(name, "filtered_keepiso"))
    assert np.array_equal(morph.filter_branch_seg_mask(unpack(name, "mask")), unpack(name, "filtered"))


@pytest.mark.parametrize("name", NAMES)
def test_medial_axis(name):
    filt = unpack(name, "filtered")
    sk, dist = morph.medial_axis(filt)
    assert np.array_equal(sk, unpack(name, "ma_skel"))
    sha = np.frombuffer(hashlib.sha256(np.ascontiguousarray(dist).tobytes()).digest(), np.uint8)
    assert np.array_equal(sha, G[name + "_ma_dist_sha"])


# ---- cv2.resize(INTER_LANCZOS4) on uint8: fixed-point path, pinned to hand-derived vectors (OpenCV is absent) -------
def _lanczos_ref(x):
    """the textbook Lanczos-4 kernel sinc(t) sinc(t / 4) at the 8 tap distances of fractional offset x, normalised, in
    extended precision -- an evaluation that shares no code with oracle/morph.py (which restates OpenCV's trig-identity
    form in float32)"""
    from fractions import Fraction
    import math
    t = [x + 3 - i for i in range(8)]
    w = [1.0 if abs(v) < 1e-12 else (math.sin(math.pi * v) / (math.pi * v)) * (math.sin(math.pi * v / 4) / (math.pi * v / 4)) for v in t]
    s = math.fsum(w)
    return [v / s for v in w]


def test_lanczos_u8_identity_and_constant():
    rs = np.random.RandomState(2)
    img = rs.randint(0, 256, (24, 40)).astype(np.uint8)
    # same size: every fractional offset is 0, the taps are (0,0,0,2048,0,0,0,0) in both passes: (v << 22 + 2^21) >> 22 = v
    assert np.array_equal(morph.lanczos4_resize_u8(img, img.shape), img)
    # a constant image: every output is (c * sum(alpha) * sum(beta) + 2^21) >> 22 with the ROUNDED 11-bit taps, whose sums
    # are 2047..2049 (cv2 does not renormalise them), so 255 may come out as 254 / 255 but never leaves that band
    c = np.full((16, 16), 255, np.uint8)
    out = morph.lanczos4_resize_u8(c, (10, 10))
    assert out.min() >= 254 and out.max() <= 255


def test_lanczos_u8_hand_derived_downscale():
    """16 -> 10 samples (the reference's 0.625): dst d reads src around fx = (d + 0.5) * 1.6 - 0.5.  Expected values from
    the textbook kernel, rounded to 11 bits by hand here; rows where a textbook tap * 2048 lies within 0.02 of a rounding
    boundary are skipped (float32 evaluation order could legitimately flip them)."""
    import math
    n_src, n_dst = 16, 10
    rs = np.random.RandomState(7)
    row = rs.randint(0, 256, n_src).astype(np.int64)
    img = np.tile(row[None, :].astype(np.uint8), (n_src, 1))           # constant along y: vertical pass = identity taps? no: also resampled
    got = morph.lanczos4_resize_u8(img, (n_src, n_dst))                 # keep the height: vertical taps are (…,2048,…)
    checked = 0
    for d in range(n_dst):
        fx = (d + 0.5) * (n_src / n_dst) - 0.5
        sx = math.floor(fx)
        w = _lanczos_ref(fx - sx)
        if any(abs((v * 2048) % 1 - 0.5) < 0.02 for v in w):
            continue
        a = [round(v * 2048) for v in w]
        acc = sum(int(row[min(max(sx - 3 + k, 0), n_src - 1)]) * a[k] for k in range(8))
        want = min(max((acc * 2048 + (1 << 21)) >> 22, 0), 255)
        assert int(got[3, d]) == want, (d, got[3, d], want)
        checked += 1
    assert checked >= 6
