"""GPU parity: the Z-stack (Sato) branch (reference scripts/compute_branches.py:224-306) through the C-ABI against
oracle/sato.py, scipy.ndimage itself and the scikit-image 0.18.3 goldens (tests/golden/sato.npz).  Float stages are
bit-exact (the host hands scipy's own gaussian tables to the library, tmat_amd/sato.py:install_gaussian_tables); masks
are equal; the result row is equal."""
from pathlib import Path

import numpy as np
import pytest
from scipy import ndimage as ndi

pytestmark = pytest.mark.gpu

G = np.load(Path(__file__).parent / "golden" / "sato.npz")


@pytest.fixture(scope="module")
def plain():
    from tmat_amd import _lib
    h = _lib.Handle(None, 0)
    yield h
    h.close()


@pytest.fixture(scope="module")
def inputs():
    from make_goldens import sato_inputs
    return sato_inputs()


def same_bits(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("sigma,mode,name", [(1.0, 0, "nearest"), (2.0, 1, "reflect"), (3.7, 2, "mirror"), (15.0, 1, "reflect")])
def test_gaussian_f32_equals_scipy(plain, inputs, sigma, mode, name):
    from tmat_amd import sato
    im = inputs[0]["t1"]
    assert same_bits(sato.gaussian(plain, im, sigma, mode), ndi.gaussian_filter(im, sigma, mode=name))
    vol = inputs[1]                                     # 3-D: Z (5 slices, shorter than the kernel), rows, columns
    assert same_bits(sato.gaussian(plain, vol, sigma, mode), ndi.gaussian_filter(vol, sigma, mode=name))


def test_gaussian_matches_skimage_golden(plain, inputs):
    from tmat_amd import sato
    for k in ("t1", "t2"):
        assert same_bits(sato.gaussian(plain, inputs[0][k], 1.0, sato.NEAREST), G[k + "_gauss"])


@pytest.mark.parametrize("k", ["t1", "t2"])
def test_sato_gradient_form_matches_skimage_0_18_goldens(plain, inputs, k):
    from tmat_amd import sato
    im = inputs[0][k]
    assert same_bits(sato.sato(plain, im, hessian="gradient"), G[k + "_sato"])
    assert same_bits(sato.sato(plain, im, (1, 3, 5), hessian="gradient"), G[k + "_sato135"])


def test_sato_derivative_form_equals_the_scipy_composition(plain, inputs):
    """scikit-image >= 0.20's Hessian: every filter is the real scipy.ndimage.gaussian_filter(order=...) in the oracle"""
    from oracle import sato as osato
    from tmat_amd import sato
    batch = np.stack([inputs[0]["t2"], inputs[0]["t2"][::-1].copy(), inputs[0]["t2"].T.copy()])
    got = sato.sato(plain, batch)
    for i in range(3):
        assert same_bits(got[i], osato.sato2d_derivatives(batch[i]))
    assert same_bits(sato.sato(plain, inputs[0]["t1"], (1, 2.5, 4)), osato.sato2d_derivatives(inputs[0]["t1"], (1, 2.5, 4)))


@pytest.mark.parametrize("shape,out_hw,bits", [((5, 210, 260), (62, 77), 16), ((3, 96, 100), (96, 100), 16), ((4, 300, 128), (150, 64), 8),
                                             ((2, 64, 40), (128, 80), 16)])
def test_stack_prepare(plain, shape, out_hw, bits):
    """per-slice gaussian + truncation, anti-aliased resize (down, none, factor 2, up), global clip and rescale"""
    from oracle import sato as osato
    from tmat_amd import sato
    rs = np.random.RandomState(shape[1])
    stack = ndi.gaussian_filter(rs.uniform(0, 1, shape) ** 4, (0, 2, 2))
    stack = (stack / stack.max() * (2 ** bits - 1)).astype(np.uint16 if bits == 16 else np.uint8)
    assert same_bits(sato.stack_prepare(plain, stack, out_hw), osato.stack_prepare(stack, out_hw))


def test_stack_prepare_constant_stack(plain):
    from oracle import sato as osato
    from tmat_amd import sato
    stack = np.full((3, 50, 60), 1234, np.uint16)
    assert same_bits(sato.stack_prepare(plain, stack, (25, 30)), osato.stack_prepare(stack, (25, 30)))


def _synth_vol(Z, h, w, seed):
    from oracle import sato as osato
    from tmat_amd import synth
    stack = synth.synth_stack(seed, Z, h * 2, w * 2, n_vessels=10)
    return stack, osato.stack_prepare(stack, (h, w))


@pytest.mark.parametrize("hessian", ["gradient", "gaussian_derivatives"])
def test_vessel_field_stage_by_stage(plain, hessian):
    from oracle import sato as osato
    from tmat_amd import sato
    _, vol = _synth_vol(6, 120, 160, 3)
    field, st = sato.vessel_field(plain, vol, hessian, return_stages=True)
    ofield, ost = osato.vessel_field(vol, return_stages=True, hessian=hessian)
    for k in ("vess", "sharp", "vessels"):
        assert same_bits(st[k], ost[k]), k
    for k in ("edges", "skel", "mask_sel", "grown", "closed", "filt"):
        assert np.array_equal(st[k], ost[k]), k
    assert ost["edges"].sum() > 100 and ost["filt"].sum() > 100           # the case exercises every stage
    assert same_bits(field, ofield)


def test_vessel_field_golden_volume(plain, inputs):
    """the 5-slice golden volume: the sharpened volume and the canny edges of scikit-image 0.18.3 itself (sigmas 1, 2, 3 there;
    here through the stage outputs of the full sigma set, so only the shared machinery is compared with the oracle)"""
    from oracle import sato as osato
    from tmat_amd import sato
    vol = inputs[1]
    field, st = sato.vessel_field(plain, vol, "gradient", return_stages=True)
    ofield, ost = osato.vessel_field(vol, return_stages=True, hessian="gradient")
    assert same_bits(st["sharp"], ost["sharp"]) and np.array_equal(st["edges"], ost["edges"]) and same_bits(field, ofield)


@pytest.mark.parametrize("hessian", ["gradient", "gaussian_derivatives"])
def test_analyze_stack_row_equals_oracle(plain, hessian):
    from oracle import pipeline, sato as osato
    from tmat_amd import branches, sato, synth
    stack = synth.synth_stack(7, 5, 300, 400, n_vessels=12)
    cfg = {"graph_thresh_1": 5, "graph_thresh_2": 10, "graph_smoothing_window": 12, "min_branch_length": 12, "remove_isolated_branches": False}
    width_um = 1000.0
    sw, mn, mx = branches.graph_px_params(cfg, 384, width_um)
    n, tot, avg, field = sato.analyze_stack(plain, stack, 5, 10, sw, mn, mx, False, hessian=hessian, return_field=True)
    on, otot, oavg = osato.analyze_stack(stack, cfg, width_um, hessian=hessian)
    assert field.shape == (288, 384)
    assert (n, tot, avg) == (on, otot, oavg)
    assert n > 0


def _well_stack(seed, Z=4, h=300, w=400):
    """a synthetic stack whose vessels sit inside a round well that is brighter than its surroundings"""
    from tmat_amd import synth
    st = synth.synth_stack(seed, Z, h, w, n_vessels=14).astype(np.float64)
    yy, xx = np.mgrid[0:h, 0:w]
    inside = (xx - w * 0.5) ** 2 / (w * 0.42) ** 2 + (yy - h * 0.52) ** 2 / (h * 0.42) ** 2 < 1
    out = np.where(inside, st * 0.6 + 14000.0, st * 0.05 + 900.0)
    return np.clip(out, 0, 65535).astype(np.uint16)


@pytest.mark.parametrize("shape,out_hw,bits", [((210, 260), (62, 77), 16), ((96, 100), (96, 100), 16), ((300, 128), (150, 64), 8)])
def test_resize_aa_of_a_projection(plain, shape, out_hw, bits):
    """tmat_resize_aa_u16 (compute_branches.py:232-238: resize of the max projection, float64 out) against the two scipy calls"""
    from oracle import sato as osato
    from tmat_amd import sato
    rs = np.random.RandomState(shape[0])
    img = ndi.gaussian_filter(rs.uniform(0, 1, shape) ** 4, 2)
    img = (img / img.max() * (2 ** bits - 1)).astype(np.uint16 if bits == 16 else np.uint8)
    got = sato.resize_aa(plain, img, out_hw)
    want = osato.resize_aa(img, out_hw)
    assert got.dtype == np.float64 and np.array_equal(got.view(np.uint64), want.view(np.uint64))


def test_stack_detect_well_prunes_the_graph_like_the_oracle(plain):
    """--detect-well on a Z stack (compute_branches.py:227-243, 412-420): masks equal to the oracle's (whose well detection is pinned to
    the reference), and the pruned row equals the oracle's and differs from the unpruned one"""
    from oracle import sato as osato
    from tmat_amd import branches, sato
    stack = _well_stack(11)
    cfg = {"graph_thresh_1": 5, "graph_thresh_2": 10, "graph_smoothing_window": 12, "min_branch_length": 12, "remove_isolated_branches": False}
    width_um = 1000.0
    sw, mn, mx = branches.graph_px_params(cfg, 384, width_um)
    field = sato.stack_field(plain, stack)
    well, pruning = sato.stack_well_masks(plain, stack, field.shape, seed=3)
    (on, otot, oavg), owell, opruning = osato.analyze_stack(stack, cfg, width_um, detect_well=True, well_seed=3, return_masks=True)
    assert 0.4 < owell.mean() < 0.95, "the test stack should have a real well"
    assert np.array_equal(well, owell) and np.array_equal(pruning, opruning)
    row = sato.field_stats(plain, field, 5, 10, sw, mn, mx, False, pruning_mask=pruning)
    assert row == (on, otot, oavg) and on > 0
    plain_row = sato.field_stats(plain, field, 5, 10, sw, mn, mx, False)
    assert plain_row == osato.analyze_stack(stack, cfg, width_um)
    assert plain_row != row, "the pruning mask should remove something in this case"


def test_pooled_workspaces_hold_nothing_a_call_reads_before_writing(plain):
    """the Z-stack calls take their device blocks from a pool on the handle (blocks of earlier calls, stale contents): with every pooled
    block filled with 0xFF (NaN / -1) or 0x7F between calls, the same stack and a stack of another geometry give the rows of a fresh
    handle -- no call reads a location of a recycled block that it has not written"""
    from tmat_amd import _lib, branches, sato, synth
    cfg = {"graph_thresh_1": 5, "graph_thresh_2": 10, "graph_smoothing_window": 12, "min_branch_length": 12}
    sw, mn, mx = branches.graph_px_params(cfg, 384, 1000.0)
    a = synth.synth_stack(11, 5, 300, 300, n_vessels=12)
    b = synth.synth_stack(12, 3, 256, 320, n_vessels=10)
    fresh = _lib.Handle(None, 0)
    try:
        want = [sato.analyze_stack(fresh, s, 5, 10, sw, mn, mx, False) for s in (a, b)]
    finally:
        fresh.close()
    for pattern in (0xFF, 0x7F):
        for s, w in ((a, want[0]), (b, want[1]), (a, want[0])):
            plain.debug_poison(pattern)
            assert sato.analyze_stack(plain, s, 5, 10, sw, mn, mx, False) == w


def test_bad_arguments(plain):
    from tmat_amd import _lib, sato
    with pytest.raises(ValueError):
        sato.stack_prepare(plain, np.zeros((2, 8, 8), np.float32), (4, 4))
    with pytest.raises(_lib.TmatError):
        sato.vessel_field(plain, np.zeros((1, 16, 16), np.float32))          # a single slice has no slice pair
    w = np.ones(3)
    assert _lib.lib().tmat_set_gaussian_table(plain.raw, -1.0, 0, 1, _lib.ptr(w)) != 0


def test_two_slice_stack_and_odd_geometry(plain):
    """Z = 2 (one slice pair: the volume's Z axis has length 1 in the unsharp mask's 3-D gaussian) on a field whose sides are not
    multiples of the kernels' tile sizes"""
    from oracle import sato as osato
    from tmat_amd import sato, synth
    stack = synth.synth_stack(21, 2, 2 * 75, 2 * 101, n_vessels=6)
    vol = osato.stack_prepare(stack, (75, 101))
    field, st = sato.vessel_field(plain, vol, "gaussian_derivatives", return_stages=True)
    ofield, ost = osato.vessel_field(vol, return_stages=True, hessian="gaussian_derivatives")
    assert same_bits(st["sharp"], ost["sharp"]) and np.array_equal(st["edges"], ost["edges"]) and np.array_equal(st["filt"], ost["filt"])
    assert same_bits(field, ofield)


def test_blank_stack_gives_an_empty_result(plain):
    """no signal at all: no edges, an all-zero field, no branches (the reference prints "No branches found" and writes no row
    for such an image; here the row carries a count of 0)"""
    from tmat_amd import sato
    stack = np.full((3, 128, 160), 1000, np.uint16)
    n, tot, avg, field = sato.analyze_stack(plain, stack, 5, 10, 5, 5, None, False, return_field=True)
    assert n == 0 and tot == 0 and not field.any()
