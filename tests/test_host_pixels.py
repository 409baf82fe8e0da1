"""CPU: the product's host pixel stages (csrc/postproc.cpp through the C-ABI) against oracle/morph.py,
bit-exact, plus oracle self-checks against scipy."""
from pathlib import Path

import numpy as np
import pytest
from scipy import ndimage as ndi

from oracle import morph
from tmat_amd import _lib, synth

G = np.load(Path(__file__).parent / "golden" / "filter.npz")
NAMES = ["d5", "m1", "blobs", "noise", "small", "empty", "full"]


def unpack(name, key):
    shape = tuple(G[name + "_shape"])
    return np.unpackbits(G[name + "_" + key])[: shape[0] * shape[1]].reshape(shape).astype(bool)


def test_permutation_matches_numpy_randomstate():
    for n in (1, 2, 7, 1000, 70001):
        assert np.array_equal(_lib.host_permutation(0, n), np.random.RandomState(0).permutation(np.arange(n)))
    assert np.array_equal(_lib.host_permutation(5, 333), np.random.RandomState(5).permutation(np.arange(333)))


@pytest.mark.parametrize("name", NAMES)
def test_filter_skeleton_medial_axis(name):
    m = unpack(name, "mask")
    assert np.array_equal(_lib.host_skeletonize(unpack(name, "median")), unpack(name, "skel"))
    assert np.array_equal(_lib.host_filter_mask(m), unpack(name, "filtered"))
    assert np.array_equal(_lib.host_filter_mask(unpack(name, "median"), False, False), unpack(name, "filtered_keepiso"))
    filt = unpack(name, "filtered")
    sk, dist = _lib.host_medial_axis(filt)
    osk, odist = morph.medial_axis(filt)
    assert np.array_equal(sk, osk) and np.array_equal(sk, unpack(name, "ma_skel"))
    assert np.array_equal(dist.view(np.uint64), odist.view(np.uint64))


def test_edt_without_background_matches_scipy_quirk():
    a = np.ones((7, 9), bool)
    _, dist = _lib.host_medial_axis(a)
    assert np.array_equal(dist, ndi.distance_transform_edt(a))


@pytest.mark.parametrize("shape,out", [((640, 640), (384, 384)), ((400, 640), (240, 384)), ((100, 130), (77, 91))])
def test_oracle_resize_equals_scipy(shape, out):
    rs = np.random.RandomState(1)
    a = rs.uniform(0, 1, shape) * (rs.uniform(size=shape) > 0.3)
    assert np.array_equal(morph.resize_aa(a, out), morph.resize_aa_scipy(a, out))


@pytest.mark.parametrize("size", [256, 640])
def test_preprocess_lanczos_rescale(size):
    img = synth.synth_image(3, size)
    tgt = morph.target_shape(img.shape, 0.625)
    a = _lib.host_lanczos4_u16(img, tgt)
    b = morph.lanczos4_resize_u16(img, tgt)
    assert np.array_equal(a, b)
    x = _lib.host_rescale01_u16(a)
    y = morph.rescale_intensity(b, (0, 1)).astype(np.float32)
    assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    # resampling sanity: a constant image stays constant, a ramp stays monotone
    c = np.full((64, 80), 1234, np.uint16)
    assert np.all(_lib.host_lanczos4_u16(c, (40, 50)) == 1234)


def test_lanczos_nonsquare_and_extremes():
    rs = np.random.RandomState(2)
    img = rs.randint(0, 65536, (97, 131)).astype(np.uint16)
    tgt = morph.target_shape(img.shape, 0.625)
    assert np.array_equal(_lib.host_lanczos4_u16(img, tgt), morph.lanczos4_resize_u16(img, tgt))


@pytest.mark.parametrize("name", ["d5", "blobs", "small", "empty"])
def test_postprocess_field_bitexact(name):
    m = unpack(name, "mask")
    rs = np.random.RandomState(9)
    # a probability map consistent with the mask: high inside, low outside, smooth
    pred = ndi.gaussian_filter(m.astype(np.float64), 1.5) * 0.98 + rs.uniform(0, 0.01, m.shape)
    out_shape = morph.dsamp_shape(m.shape, 384) if name == "d5" else (m.shape[0] * 3 // 5, m.shape[1] * 3 // 5)
    ref, seg, skel = morph.postprocess(pred, out_shape)
    got = _lib.host_postprocess(pred, out_shape)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    r0 = morph.rescale_intensity(ref, (0, 255))
    assert r0.dtype == np.float32
    assert np.array_equal(_lib.host_rescale255_f32(got).view(np.uint32), r0.view(np.uint32))
