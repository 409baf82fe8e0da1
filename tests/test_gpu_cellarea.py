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
    r = subprocess.run([sys.executable, str(script), str(ind), str(outd), "-w"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 1 and "detect-well" in r.stdout
