"""GPU parity: the cell-area tool (reference scripts/compute_cell_area.py, preprocessing.exec_threshold) through
tmat_cell_area_batch against oracle/cellarea.py (same deterministic algorithm: equal masks, parameters to 1e-9) and against the
reference's own exec_threshold goldens (scikit-learn inside: 0.1 percentage points of the area, see test_oracle_cellarea.py)."""
import csv
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parents[1]
G = np.load(Path(__file__).parent / "golden" / "cellarea.npz")
AREA_TOL = 0.1 / 100.0


@pytest.fixture(scope="module")
def plain():
    from tmat_amd import _lib
    h = _lib.Handle(None, 0)
    yield h
    h.close()


def test_batch_equals_oracle(plain):
    from oracle import cellarea as ca
    from tmat_amd import preprocessing, synth
    imgs = np.stack([synth.synth_image(20 + i, 640, n_vessels=25 + 5 * i) for i in range(3)])
    for sd in (0.0, -0.5):
        area, kept, params = preprocessing.cell_area_batch(plain, imgs, 256, sd, return_params=True)
        for i in range(3):
            small = ca.resize_linear_u16(imgs[i], ca.resized_shape(imgs[i].shape, 256))
            x = ca.rescale01(small)
            thresh, okept = ca.gmm_threshold(x, sd, levels=small)
            assert params[i, 0] == pytest.approx(thresh, rel=1e-9)
            assert np.array_equal(kept[i] > 0, okept) and area[i] == okept.sum() / okept.size
            assert params[i, 8] == 1.0 and 0.02 < area[i] < 0.7


def test_resize_is_bit_exact_and_nonsquare_swaps(plain):
    from oracle import cellarea as ca
    from tmat_amd import preprocessing
    rs = np.random.RandomState(3)
    img = (rs.uniform(0, 1, (300, 500)) ** 3 * 65535).astype(np.uint16)
    oh, ow = preprocessing.resized_shape(img.shape, 128)
    assert (oh, ow) == ca.resized_shape(img.shape, 128) == (128, 77)
    area, kept = preprocessing.cell_area_batch(plain, img[None], 128, 0.0)
    oa, ok = ca.cell_area(img, 128, 0.0)
    assert kept.shape == (1, 128, 77) and np.array_equal(kept[0], ok) and area[0] == oa


@pytest.mark.parametrize("k", ["c0", "c1", "c2"])
@pytest.mark.parametrize("sd", [0.0, 0.5])
def test_against_the_reference_function_goldens(plain, k, sd):
    from make_goldens import cellarea_inputs
    from tmat_amd import preprocessing
    img = cellarea_inputs()[k]
    area, kept = preprocessing.cell_area_batch(plain, img[None], None, sd)
    want = np.unpackbits(G[f"{k}_sd{sd}_bits"])[: img.size].reshape(img.shape).astype(bool)
    assert abs(area[0] - float(G[f"{k}_sd{sd}_area"])) <= AREA_TOL
    assert ((kept[0] > 0) != want).mean() <= AREA_TOL


def test_degenerate_images(plain):
    from oracle import cellarea as ca
    from tmat_amd import preprocessing
    flat = np.full((1, 64, 64), 700, np.uint16)                       # one level: the mixture collapses onto it
    two = np.zeros((1, 64, 64), np.uint16); two[0, :, 40:] = 900      # two levels
    for im in (flat, two):
        area, kept = preprocessing.cell_area_batch(plain, im, None, 0.0)
        oa, ok = ca.cell_area(im[0], None, 0.0)
        assert area[0] == oa and np.array_equal(kept[0], ok)
    assert preprocessing.cell_area_batch(plain, np.zeros((0, 8, 8), np.uint16), None)[0].shape == (0,)


def test_script_end_to_end(tmp_path):
    from PIL import Image
    from oracle import cellarea as ca
    from tmat_amd import synth
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    imgs = {f"w{i}": synth.synth_image(30 + i, 300, n_vessels=12, scale=0.5) for i in range(3)}
    for k, v in imgs.items():
        Image.fromarray(v).save(ind / f"{k}.tif")
    script = REPO / "tissue-model-analysis-tools_amd" / "scripts" / "compute_cell_area.py"
    r = subprocess.run([sys.executable, str(script), str(ind), str(outd), "--sd-coef", "0.25"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = list(csv.reader(open(outd / "calculations" / "cell_area.csv")))
    assert rows[0] == ["image_id", "area_pct"] and sorted(x[0] for x in rows[1:]) == sorted(imgs)
    for name, pct in rows[1:]:
        oa, ok = ca.cell_area(imgs[name], 512, 0.25)
        assert float(pct) == oa * 100
        assert np.array_equal(np.array(Image.open(outd / "thresholded" / f"{name}_thresholded.png")), ok)
    # --detect-well: the same run with well masks (a second CSV next to the first: get_unique_output_filepath)
    r = subprocess.run([sys.executable, str(script), str(ind), str(outd), "--sd-coef", "0.25", "-w", "--well-seed", "3"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = list(csv.reader(open(outd / "calculations" / "cell_area-2.csv")))
    for name, pct in rows[1:]:
        oa, ok, ow_ = ca.cell_area_well(imgs[name], 512, 0.25, seed=3)
        assert float(pct) == oa * 100
        assert np.array_equal(np.array(Image.open(outd / "thresholded" / f"{name}_thresholded-2.png")), ok)
        assert np.array_equal(np.array(Image.open(outd / "thresholded" / f"{name}_well_mask.png")), ow_)


def _well_plate(seed, shape=(600, 640), dtype=np.uint16):
    """cells (bright blobs) inside a round well that is brighter than its surroundings"""
    from scipy import ndimage as ndi
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
    inside = (xx - shape[1] * 0.5) ** 2 + (yy - shape[0] * 0.52) ** 2 < (min(shape) * 0.43) ** 2
    a = np.where(inside, 0.35, 0.05) + rs.normal(0, 0.01, shape)
    blobs = ndi.gaussian_filter((rs.uniform(size=shape) < 0.002).astype(float), 4) * 120
    a = a + np.where(inside, blobs, 0)
    top = 60000 if dtype == np.uint16 else 250
    return np.clip(a / a.max() * top, 0, top).astype(dtype)


@pytest.mark.parametrize("dtype", [np.uint16, np.uint8])
def test_detect_well_form_equals_the_oracle(dtype):
    """--detect-well (compute_cell_area.py:117-130, 273-286): well mask of the down-sampled image, mixture inside it, area
    relative to the well; equal to oracle/cellarea.py:cell_area_well, whose well mask is pinned to the reference"""
    from oracle import cellarea as oc
    from tmat_amd import _lib, preprocessing
    h = _lib.Handle(None, 0)
    try:
        imgs = np.stack([_well_plate(1, dtype=dtype), _well_plate(2, dtype=dtype)])
        area, kept, well = preprocessing.cell_area_batch_well(h, imgs, 512, 0.0, well_seed=7)
    finally:
        h.close()
    for i in range(len(imgs)):
        a0, k0, w0 = oc.cell_area_well(imgs[i], 512, 0.0, seed=7)
        assert 0.3 < (w0 > 0).mean() < 0.9, "the test image should have a real well"
        assert np.array_equal(well[i], w0)
        assert np.array_equal(kept[i], k0)
        assert area[i] == a0 and 0 < a0 < 1


def test_exact_halving_takes_cv2s_area_path():
    """cv2.resize(INTER_LINEAR) with both scale factors exactly 2 = INTER_AREA's (a + b + c + d + 2) >> 2 (resize.cpp): hand-derived
    vectors, a 2 x 2 block summing to 4 k + 2 rounds UP (bilinear + round-half-even would round to even)"""
    from oracle import cellarea as oc
    from tmat_amd import _lib, preprocessing
    img = np.array([[1, 2, 10, 10], [2, 1, 10, 11], [0, 0, 65535, 65535], [0, 1, 65535, 65534]], np.uint16)
    want = np.array([[2, 10], [0, 65535]], np.uint16)           # (6 + 2) >> 2 = 2, (41 + 2) >> 2 = 10, (1 + 2) >> 2 = 0, (262139 + 2) >> 2 = 65535
    assert np.array_equal(oc.resize_linear_u16(img, (2, 2)), want)
    big = np.random.RandomState(0).randint(0, 65536, (3, 64, 64)).astype(np.uint16)
    h = _lib.Handle(None, 0)
    try:
        got = preprocessing.resize_batch(h, big, 32)             # 64 x 64 -> 32 x 32: exactly 2 on both axes
        odd = preprocessing.resize_batch(h, big, 40)             # not a halving: the bilinear path
    finally:
        h.close()
    for i in range(3):
        a = big[i].astype(np.uint32)
        assert np.array_equal(got[i], ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint16))
        assert np.array_equal(got[i], oc.resize_linear_u16(big[i], (32, 32)))
        assert np.array_equal(odd[i], oc.resize_linear_u16(big[i], oc.resized_shape((64, 64), 40)))


def test_eight_bit_sources_take_cv2s_fixed_point_bilinear():
    """uint8 images: cv2.resize(INTER_LINEAR) is the fixed-point path (11-bit coefficients, (b0 (S0 >> 4) >> 16) + (b1 (S1 >> 4) >> 16) + 2 >> 2),
    not the float path of uint16 images (oracle/cellarea.py:resize_linear_u8, hand-derived vectors in tests/test_oracle_cellarea.py);
    reference call site scripts/compute_cell_area.py:54-57.  The device follows the source dtype; the whole tool on uint8 images equals
    the oracle's, and the two arithmetic paths do differ on the same pixels."""
    from oracle import cellarea as oc
    from tmat_amd import _lib, preprocessing
    rs = np.random.RandomState(12)
    imgs8 = rs.randint(0, 256, (3, 90, 120)).astype(np.uint8)
    h = _lib.Handle(None, 0)
    try:
        got8 = preprocessing.resize_batch(h, imgs8, 77)                         # 90 x 120 -> 58 x 77
        got16 = preprocessing.resize_batch(h, imgs8.astype(np.uint16), 77)      # the same pixels as uint16: float path
        plate = np.stack([_well_plate(3, (300, 320), np.uint8), _well_plate(4, (300, 320), np.uint8)])
        area, kept = preprocessing.cell_area_batch(h, plate, 200, 0.0)
        same = preprocessing.resize_batch(h, imgs8.astype(np.uint16), 77)       # the depth is per call: back on the float path
    finally:
        h.close()
    shp = oc.resized_shape((90, 120), 77)
    for i in range(3):
        assert np.array_equal(got8[i], oc.resize_linear_u8(imgs8[i], shp))
        assert np.array_equal(got16[i], oc.resize_linear_u16(imgs8[i].astype(np.uint16), shp))
    assert (got8 != got16).any() and np.abs(got8.astype(int) - got16.astype(int)).max() == 1
    assert np.array_equal(same, got16)
    for i in range(2):
        a0, k0 = oc.cell_area(plate[i], 200, 0.0)
        assert area[i] == a0 and np.array_equal(kept[i], k0)
