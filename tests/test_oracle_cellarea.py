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


def test_cell_area_of_a_stack_uses_the_max_projection():
    rs = np.random.RandomState(1)
    st = (rs.uniform(0, 1, (3, 40, 40)) ** 6 * 60000).astype(np.uint16)
    a1, m1 = ca.cell_area(st, None)
    a2, m2 = ca.cell_area(st.max(0), None)
    assert a1 == a2 and np.array_equal(m1, m2) and set(np.unique(m1)) <= {0, 255}
