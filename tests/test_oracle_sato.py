"""CPU: oracle/sato.py (the Z-stack / Sato branch, compute_branches.py:224-306) against goldens that scikit-image 0.18.3
produced stage by stage (tools/make_goldens.py sato, /opt/conda/bin/python3.9).  Filters that are pure float32 / float64
arithmetic in a fixed order (sato, gaussian, unsharp mask) must match to the last bit; masks must be equal."""
from pathlib import Path

import numpy as np
import pytest

from make_goldens import sato_inputs
from oracle import morph, sato

G = np.load(Path(__file__).parent / "golden" / "sato.npz")
IMGS, VOL = sato_inputs()


def bits(name, shape):
    return np.unpackbits(G[name])[: shape[0] * shape[1]].reshape(shape).astype(bool)


@pytest.mark.parametrize("k", ["t1", "t2"])
def test_sato_and_gaussian_bit_exact(k):
    im = IMGS[k]
    assert np.array_equal(sato.sato2d(im), G[k + "_sato"])
    assert np.array_equal(sato.sato2d(im, (1, 3, 5)), G[k + "_sato135"])
    g = sato.gaussian(im)
    assert g.dtype == G[k + "_gauss"].dtype and np.array_equal(g, G[k + "_gauss"])


def _vess():
    return np.stack([sato.sato2d(np.maximum(VOL[z], VOL[z + 1]), (1, 2, 3)) for z in range(len(VOL) - 1)])


def test_unsharp_canny_medial_regionprops():
    sharp = sato.unsharp_mask(_vess(), 2, 2)
    assert sharp.dtype == np.float32 and np.array_equal(sharp, G["vol_sharp"])
    vessels = sharp.max(0)
    edges = sato.canny0(vessels)
    assert np.array_equal(edges, bits("vol_edges", vessels.shape))
    assert np.array_equal(sato.canny0(G["t1_sato"]), bits("t1_edges", G["t1_sato"].shape))
    skel, _ = morph.medial_axis(edges)
    assert np.array_equal(skel, bits("vol_skel", vessels.shape))
    np.testing.assert_allclose(sato.ecc_times_diameter(skel), G["vol_eccdiam"], rtol=1e-12, atol=1e-12)    # LAPACK eigvalsh vs here: ulp level


def test_closing_and_dilation():
    from scipy import ndimage as ndi
    m = np.random.RandomState(3).uniform(size=(60, 70)) > 0.7
    d2 = sato.disk(2)
    closed = ndi.grey_erosion(ndi.grey_dilation(m.astype(np.uint8), footprint=d2), footprint=d2).astype(bool)
    assert np.array_equal(closed, bits("m_closed", m.shape))
    dil = ndi.grey_dilation(m.astype(np.uint8), footprint=np.ones((3, 3), bool)).astype(bool)
    assert np.array_equal(dil, bits("m_dil", m.shape))


def test_region_grow_against_a_pixel_loop():
    """compute_branches.py:283-294 is plain numpy in the script (not importable: cv2, aicsimageio): pinned here by an
    independent per-pixel evaluation of the same rule"""
    rs = np.random.RandomState(8)
    v = rs.uniform(0, 0.05, (30, 40)).astype(np.float32)
    m0 = rs.uniform(size=v.shape) > 0.9
    got = sato.region_grow(m0, v, 3)
    m = m0.copy()
    for _ in range(3):
        new = m.copy()
        for y in range(v.shape[0]):
            for x in range(v.shape[1]):
                hi = lo = False
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        yy, xx = y + dy, x + dx
                        if (dy or dx) and 0 <= yy < v.shape[0] and 0 <= xx < v.shape[1] and m[yy, xx]:
                            if v[y, x] < v[yy, xx]:
                                lo = True
                            else:
                                hi = True
                if v[y, x] > 0.01 and hi and not lo:
                    new[y, x] = True
        m = new
    assert np.array_equal(got, m)


def test_vessel_field_runs_end_to_end():
    field, st = sato.vessel_field(VOL, return_stages=True)
    assert field.dtype == np.float32 and field.shape == VOL.shape[1:] and np.isfinite(field).all()
    assert st["edges"].any() and field.max() > 0


def test_stack_prepare_3d_scipy_calls_equal_the_per_slice_form():
    """the resize of the 3-D stack (Z sigma 0, Z zoom 1) is the 2-D anti-aliased resize of every slice, clipped to the range
    of the WHOLE stack: that is what the device path computes (csrc/stack_pipeline.cpp:stack_prepare_dev)"""
    from scipy import ndimage as ndi
    rs = np.random.RandomState(5)
    stack = (rs.uniform(0, 1, (4, 70, 90)) ** 3 * 60000).astype(np.uint16)
    out_hw = (30, 38)
    vol = sato.stack_prepare(stack, out_hw)
    assert vol.dtype == np.float32 and vol.shape == (4, 30, 38) and vol.min() == 0 and vol.max() == 1
    st = stack.copy()
    for i in range(len(st)):
        st[i, :] = ndi.gaussian_filter(st[i].astype(np.float64), 1.0, mode="nearest")
    per = np.stack([morph.resize_aa_scipy(st[i], out_hw) for i in range(len(st))])     # clips per slice ...
    fz = np.divide(st.shape[1:], out_hw)
    raw = np.stack([ndi.zoom(ndi.gaussian_filter(st[i].astype(np.float64), np.maximum(0, (fz - 1) / 2), mode="mirror"),
                             [1 / f for f in fz], order=1, mode="mirror", grid_mode=True) for i in range(len(st))])
    raw = np.clip(raw, st.min(), st.max())                                              # ... the stack form clips globally
    assert np.array_equal(morph.rescale_intensity(raw, (0, 1)).astype(np.float32), vol)
    assert per.shape == raw.shape


def test_derivative_form_responds_to_ridges_like_the_gradient_form():
    """scikit-image >= 0.20's Hessian (PARITY UNPINNED as a composition, see oracle/sato.py): both forms estimate the same
    second derivative, so on smooth tubes they agree to within the discretisation error, are non-negative and peak on ridges"""
    im = IMGS["t1"]
    a = sato.sato2d(im, (2, 3, 5), "gradient")
    b = sato.sato2d(im, (2, 3, 5), "gaussian_derivatives")
    assert b.dtype == np.float32 and b.min() >= 0
    assert np.corrcoef(a.ravel(), b.ravel())[0, 1] > 0.97
    assert abs(float(b.max()) / float(a.max()) - 1) < 0.25
