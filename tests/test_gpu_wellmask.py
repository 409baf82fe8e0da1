"""GPU parity of the --detect-well stages against the reference's own results (tests/golden/wellmask.npz: the imported
well_mask_generation module, scikit-image 0.18.3) and against oracle/wellmask.py, through the C-ABI."""
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = np.load(Path(__file__).parent / "golden" / "wellmask.npz")
NAMES = [str(n) for n in GOLD["names"]]


@pytest.fixture(scope="module")
def inputs():
    import make_goldens
    return make_goldens.wellmask_inputs()


@pytest.fixture(scope="module")
def plain():
    from tmat_amd import _lib
    h = _lib.Handle(None, 0)
    yield h
    h.close()


def _unpack(bits, shape):
    return np.unpackbits(bits)[:shape[0] * shape[1]].reshape(shape).astype(bool)


@pytest.mark.parametrize("key", NAMES)
def test_well_mask_equals_the_reference(plain, inputs, key):
    from tmat_amd import well_mask_generation as wm
    name, seed = key.rsplit("_s", 1)
    img = inputs[name]
    shape = tuple(int(v) for v in GOLD[key + "_shape"])
    assert np.array_equal(wm.auto_threshold_well(img, plain), _unpack(GOLD[key + "_thresh"], shape)), "Otsu + erosion"
    res = wm.generate_well_mask(img, return_superellipse_params=True, handle=plain, seed=int(seed))
    params = GOLD[key + "_params"]
    if params.size:
        assert isinstance(res, tuple)
        assert np.array_equal(np.array(res[1:], np.float64), params)
        mask = res[0]
    else:
        assert not isinstance(res, tuple)
        mask = res
    assert np.array_equal(np.asarray(mask) > 0, _unpack(GOLD[key + "_mask"], shape))


def test_canny_of_masks_equals_the_oracle(plain):
    from oracle import wellmask as ow
    from tmat_amd import well_mask_generation as wm
    rs = np.random.RandomState(4)
    yy, xx = np.mgrid[0:150, 0:200]
    for k in range(4):
        m = ((xx - rs.randint(60, 140)) ** 2 / rs.uniform(30, 90) ** 2 + (yy - rs.randint(50, 100)) ** 2 / rs.uniform(25, 70) ** 2 < 1)
        if k == 3:
            m = rs.uniform(size=m.shape) < 0.3                          # salt noise: many short edges and ties
        assert np.array_equal(wm._border(plain, m), ow.border_of(m)), k


def test_threshold_stage_on_odd_shapes_and_inversion(plain):
    from oracle import wellmask as ow
    from tmat_amd import well_mask_generation as wm
    rs = np.random.RandomState(8)
    for shape, invert in (((37, 53), False), ((211, 97), True), ((20, 20), False)):
        yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
        a = ((xx - shape[1] / 2) ** 2 + (yy - shape[0] / 2) ** 2 < (min(shape) * 0.4) ** 2) * 0.5 + 0.2
        if invert:
            a = 1.0 - a
        img = (a + rs.normal(0, 0.05, shape)).astype(np.float32)
        assert np.array_equal(wm.auto_threshold_well(img, plain), ow.auto_threshold_well(img)), shape
    flat = np.full((40, 40), 0.5, np.float32)                            # constant image: Otsu returns the value itself
    assert np.array_equal(wm.auto_threshold_well(flat, plain), ow.auto_threshold_well(flat))


def test_make_well_mask_equals_the_oracle(plain, inputs):
    from oracle import wellmask as ow
    from tmat_amd import well_mask_generation as wm
    for name in ("round_bright", "square_bright", "blank"):
        for seed in (0, 7):
            got = wm.make_well_mask(inputs[name], handle=plain, seed=seed, warn=lambda m: None)
            ref = ow.make_well_mask(inputs[name], seed=seed)
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), (name, seed)


def _well_image(seed, size=512):
    """a synthetic projection whose vessels sit inside a bright round well"""
    from tmat_amd import synth
    img = synth.synth_image(seed, size, n_vessels=12, scale=1.0).astype(np.float64)
    yy, xx = np.mgrid[0:size, 0:size]
    inside = (xx - size * 0.52) ** 2 + (yy - size * 0.49) ** 2 < (size * 0.42) ** 2
    img = np.where(inside, img + 12000.0, 0.0)
    return np.clip(img, 0, 65535).astype(np.uint16)


def test_branch_rows_with_detect_well_equal_the_oracle(handle, weights):
    """the --detect-well form of the 2-D branch (compute_branches.py:318-337, 359-361, 425) end to end: rows equal"""
    from oracle import pipeline
    from tmat_amd import branches
    cfg = dict(graph_thresh_1=5, graph_thresh_2=10, graph_smoothing_window=12, min_branch_length=12, remove_isolated_branches=False)
    imgs = np.stack([_well_image(3), _well_image(5)])
    fields = branches.well_fields(handle, imgs, 0.625, 16, well_seed=7, warn=lambda m: None)
    rows = branches.well_rows(handle, fields, cfg, 500.0, (5.0, 10.0))
    real = []
    for i in range(len(imgs)):
        (n0, tot0, avg0), mid = pipeline.analyze_image_well(imgs[i], weights, cfg, 500.0, seed=7)
        real.append(0.4 < mid["well"].mean() < 0.95 and mid["pruning"].any())
        assert np.array_equal(fields[i][2], mid["well"]) and np.array_equal(fields[i][1], mid["pruning"])
        assert rows[i][1] == n0 and rows[i][2] == tot0 and rows[i][3] == avg0, (rows[i], (n0, tot0, avg0))
    assert sum(r[1] for r in rows) > 0
    assert real == [False, True], "image 3: coverage below 40 % -> the mask is dropped (compute_branches.py:132-139); image 5: a real well"


def test_input_norm_is_applied_in_front_of_the_smooth_prediction(handle, weights):
    """norm_mean / norm_std of the model config (models.py:636-637) on the device: bit-equal to the oracle on the normalised image"""
    from oracle import blend, unet as ou
    x = np.random.RandomState(6).uniform(0, 1, (150, 170)).astype(np.float32)
    ref = blend.predict_img_with_smooth_windowing(((x - 0.3) / 0.8).astype(np.float32), 320, 2, ou.predict_exact(weights))
    handle.set_input_norm(0.3, 0.8)
    try:
        got = handle.predict_smooth(x)
    finally:
        handle.set_input_norm(None, None)
    assert np.array_equal(got.view(np.uint64), ref.view(np.uint64))
    assert np.array_equal(handle.predict_smooth(x).view(np.uint64),
                          blend.predict_img_with_smooth_windowing(x, 320, 2, ou.predict_exact(weights)).view(np.uint64))
