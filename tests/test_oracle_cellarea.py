"""CPU: oracle/cellarea.py (the cell-area tool, reference scripts/compute_cell_area.py + preprocessing.exec_threshold).

Two pins:
  * the EM against scikit-learn's GaussianMixture itself, started from the same parameters, on float64 pixels: the same
    trajectory (iteration count) and parameters to 1e-9;
  * the whole threshold against the reference's own exec_threshold with RandomState(0) (tests/golden/cellarea.npz, made by
    tools/make_goldens.py cellarea with the scikit-learn of this interpreter).  sklearn seeds its KMeans initialisation
    from the caller's RandomState, runs in float32 on float32 pixels and stops EM on a 1e-3 change of the mean
    log-likelihood, so the fitted threshold depends on more than the data; the deterministic restatement agrees with it to
    AREA_TOL = 0.1 percentage points of the image area (observed: < 0.03), which is the tolerance of this tool's parity.
"""
from pathlib import Path

import numpy as np
import pytest

from make_goldens import cellarea_inputs
from oracle import cellarea as ca

G = np.load(Path(__file__).parent / "golden" / "cellarea.npz")
AREA_TOL = 0.1 / 100.0


def test_em_matches_sklearn_from_the_same_start():
    from sklearn.mixture import GaussianMixture
    img = cellarea_inputs()["c1"]
    x = ca.rescale01(img).astype(np.float64)
    vals, counts = np.unique(x.ravel(), return_counts=True)
    k = ca.two_means_threshold(vals, counts)
    resp0 = np.zeros((len(vals), 2)); resp0[: k + 1, 0] = 1; resp0[k + 1:, 1] = 1
    w, mu, var, it, conv = ca.em_fit(vals, counts, resp0)
    # the parameters sklearn would estimate from those responsibilities, handed over as its explicit initialisation
    c = counts.astype(np.float64)
    nk = (resp0 * c[:, None]).sum(0) + 10 * np.finfo(np.float64).eps
    m0 = (resp0 * (c * vals)[:, None]).sum(0) / nk
    v0 = (resp0 * (c[:, None] * (vals[:, None] - m0) ** 2)).sum(0) / nk + 1e-6
    gm = GaussianMixture(n_components=2, weights_init=nk / c.sum(), means_init=m0[:, None], precisions_init=(1.0 / v0)[:, None, None],
                         random_state=0).fit(x.reshape(-1, 1))
    assert conv and gm.converged_ and gm.n_iter_ == it
    np.testing.assert_allclose(gm.means_.ravel(), mu, rtol=1e-9)
    np.testing.assert_allclose(gm.covariances_.ravel(), var, rtol=1e-8)
    np.testing.assert_allclose(gm.weights_, w, rtol=1e-9)


def test_two_means_threshold_is_the_global_optimum():
    rs = np.random.RandomState(2)
    levels = np.sort(rs.choice(5000, 300, replace=False)).astype(np.float64)
    counts = rs.randint(1, 50, 300)
    k = ca.two_means_threshold(levels, counts)
    def wss(kk):
        a, b = slice(0, kk + 1), slice(kk + 1, None)
        f = lambda s: (counts[s] * (levels[s] - np.average(levels[s], weights=counts[s])) ** 2).sum()
        return f(a) + f(b)
    best = min(range(len(levels) - 1), key=wss)
    assert k == best


@pytest.mark.parametrize("k", ["c0", "c1", "c2"])
@pytest.mark.parametrize("sd", [0.0, 0.5])
def test_threshold_agrees_with_the_reference_function(k, sd):
    img = cellarea_inputs()[k]
    thresh, kept = ca.gmm_threshold(ca.rescale01(img), sd)
    want = np.unpackbits(G[f"{k}_sd{sd}_bits"])[: img.size].reshape(img.shape).astype(bool)
    area, want_area = kept.sum() / kept.size, float(G[f"{k}_sd{sd}_area"])
    assert abs(area - want_area) <= AREA_TOL
    assert (kept != want).mean() <= AREA_TOL              # the two masks are nested thresholdings of one image
    assert 0.02 < want_area < 0.6                         # the case is not degenerate


def test_resize_linear_against_scipy_map_coordinates():
    """PARITY UNPINNED vs cv2 (absent): the bilinear kernel follows the published algorithm; here it is cross-checked against
    an independent evaluation (scipy's order-1 interpolation at the same pixel-centre coordinates, nearest border)"""
    from scipy import ndimage as ndi
    rs = np.random.RandomState(4)
    img = rs.randint(0, 65535, (50, 70)).astype(np.uint16)
    oh, ow = 23, 31
    got = ca.resize_linear_u16(img, (oh, ow))
    yy = np.clip((np.arange(oh) + 0.5) * (50 / oh) - 0.5, 0, 49)
    xx = np.clip((np.arange(ow) + 0.5) * (70 / ow) - 0.5, 0, 69)
    ref = ndi.map_coordinates(img.astype(np.float64), np.meshgrid(yy, xx, indexing="ij"), order=1, mode="nearest")
    assert np.abs(got.astype(np.float64) - ref).max() <= 1.0            # float32 vs float64 weights, rounding
    assert ca.resized_shape((1000, 1500), 512) == (512, 341)            # dsize (341, 512) is read as (width, height)


def _cv2_linear_8u_scalar(img, oh, ow):
    """cv2's 8-bit INTER_LINEAR written out pixel by pixel from the published source (imgproc/src/resize.cpp: resizeGeneric_ tables,
    HResizeLinear<uchar, int, short, 2048>, the 8-bit VResizeLinear), independent of the vectorised oracle function"""
    import math
    H, W = img.shape

    def axis(n_src, n_dst):
        tab = []
        scale = n_src / n_dst
        for d in range(n_dst):
            fx = np.float32((d + 0.5) * scale - 0.5)
            sx = int(math.floor(fx))
            fx = np.float32(fx - sx)
            if sx < 0:
                sx, fx = 0, np.float32(0)
            if sx >= n_src - 1:
                sx, fx = n_src - 1, np.float32(0)
            a0 = int(np.rint(np.float32((np.float32(1) - fx) * np.float32(2048))))       # saturate_cast<short>(cbuf[k] * INTER_RESIZE_COEF_SCALE)
            a1 = int(np.rint(np.float32(fx * np.float32(2048))))
            tab.append((sx, min(sx + 1, n_src - 1), a0, a1))
        return tab
    xs, ys = axis(W, ow), axis(H, oh)
    out = np.zeros((oh, ow), np.uint8)
    for y, (r0, r1, b0, b1) in enumerate(ys):
        for x, (c0, c1, a0, a1) in enumerate(xs):
            s0 = int(img[r0, c0]) * a0 + int(img[r0, c1]) * a1
            s1 = int(img[r1, c0]) * a0 + int(img[r1, c1]) * a1
            out[y, x] = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2
    return out


def test_resize_linear_u8_is_cv2s_fixed_point_arithmetic():
    """PARITY UNPINNED vs cv2 (absent; opencv-python >= 4.9, setup.py:63).  uint8 images go through cv2's fixed-point bilinear path
    (reference call sites scripts/compute_cell_area.py:54-57, data_prep.py:36).  Pinned to (1) a hand-derived vector, (2) an
    independent pixel-by-pixel restatement of the published source, (3) the properties the arithmetic must have: identity at equal
    size, INTER_AREA's integer mean for an exact halving, at most 1 level from the float path -- and NOT equal to it everywhere."""
    img = np.array([[10, 20], [30, 250]], np.uint8)
    # 2 x 2 -> 3 x 3 (scale 2/3): source coordinates -1/6 (clamped: weights 2048, 0), 1/2 (1024, 1024), 7/6 (clamped to the last pixel).
    # horizontal sums: row 0: 20480, 30720, 40960; row 1: 61440, 286720, 512000.  Centre: ((1024 * (30720 >> 4)) >> 16) = 30,
    # ((1024 * (286720 >> 4)) >> 16) = 280, (30 + 280 + 2) >> 2 = 78; edge rows / columns reproduce the source pixels.
    want = np.array([[10, 15, 20], [20, 78, 135], [30, 140, 250]], np.uint8)
    assert np.array_equal(ca.resize_linear_u8(img, (3, 3)), want)
    rs = np.random.RandomState(8)
    big = rs.randint(0, 256, (37, 41)).astype(np.uint8)
    for oh, ow in ((23, 29), (37, 41), (50, 64), (5, 3)):
        got = ca.resize_linear_u8(big, (oh, ow))
        assert got.dtype == np.uint8 and np.array_equal(got, _cv2_linear_8u_scalar(big, oh, ow)), (oh, ow)
    assert np.array_equal(ca.resize_linear_u8(big, (37, 41)), big)                      # equal size: identity
    even = rs.randint(0, 256, (16, 24)).astype(np.uint8)
    a = even.astype(np.int64)
    assert np.array_equal(ca.resize_linear_u8(even, (8, 12)), ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8))
    f = ca.resize_linear_u16(big.astype(np.uint16), (23, 29)).astype(np.int64)          # what the float path would give
    d = ca.resize_linear_u8(big, (23, 29)).astype(np.int64) - f
    assert np.abs(d).max() == 1 and (d != 0).mean() < 0.2                              # the two paths differ, by one level, at a few pixels
    assert ca.resize_linear(big, (23, 29)).dtype == np.uint8 and ca.resize_linear(big.astype(np.uint16), (23, 29)).dtype == np.uint16


def test_cell_area_of_a_stack_uses_the_max_projection():
    rs = np.random.RandomState(1)
    st = (rs.uniform(0, 1, (3, 40, 40)) ** 6 * 60000).astype(np.uint16)
    a1, m1 = ca.cell_area(st, None)
    a2, m2 = ca.cell_area(st.max(0), None)
    assert a1 == a2 and np.array_equal(m1, m2) and set(np.unique(m1)) <= {0, 255}
