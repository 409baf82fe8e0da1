"""GPU, BASELINE.json full size (1024x1024 uint16): one image end to end against the oracle (bit-exact
rows), and size-independent properties of the batch pipeline (order / pass-boundary invariance,
run-to-run determinism, D4 symmetry of the smooth prediction)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CFG = dict(graph_thresh_1=5, graph_thresh_2=10, graph_smoothing_window=12, min_branch_length=12,
           remove_isolated_branches=False)


@pytest.fixture(scope="module")
def big_handle(weights):
    from tmat_amd import _lib, synth
    h = _lib.Handle(synth.pack_weights(weights), 0, 600)      # 3 images per pass
    yield h
    h.close()


@pytest.fixture(scope="module")
def images():
    from tmat_amd import synth
    return np.stack([synth.synth_image(i, 1024) for i in range(4)])


def test_full_size_image_matches_oracle(big_handle, weights, images):
    from oracle import pipeline
    from tmat_amd import branches
    row = branches.analyze_batch(big_handle, images[:1], CFG, 1000.0)[0]
    n0, tot0, avg0 = pipeline.analyze_image(images[0], weights, CFG, 1000.0)
    assert n0 > 0
    assert row[1] == n0 and row[2] == tot0 and row[3] == avg0, (row, (n0, tot0, avg0))


def test_batch_order_and_pass_boundaries_do_not_matter(big_handle, images):
    from tmat_amd import branches
    a = branches.analyze_batch(big_handle, images, CFG, 1000.0)                       # passes of 3 + 1
    perm = [2, 0, 3, 1]
    b = branches.analyze_batch(big_handle, images[perm], CFG, 1000.0)
    for j, i in enumerate(perm):
        assert a[i][1:] == b[j][1:]
    seven = np.concatenate([images, images[:3]])                                      # 7 images: passes 3 + 3 + 1
    c = branches.analyze_batch(big_handle, seven, CFG, 1000.0, first_index=100)
    assert [r[0] for r in c] == list(range(100, 107))
    for j in range(7):
        assert c[j][1:] == a[j % 4][1:]
    assert a == branches.analyze_batch(big_handle, images, CFG, 1000.0)               # deterministic


def test_smooth_prediction_is_d4_consistent(big_handle):
    """predicting a flipped / rotated image equals flipping / rotating the prediction of the original: the 8
    D4 orientations are averaged in a fixed order, so this holds up to the f64 summation order (1e-12), not bitwise."""
    rs = np.random.RandomState(8)
    x = rs.uniform(0, 1, (640, 640)).astype(np.float32)
    p = big_handle.predict_smooth(x)
    q = big_handle.predict_smooth(np.ascontiguousarray(x[:, ::-1]))[:, ::-1]
    r = np.rot90(big_handle.predict_smooth(np.ascontiguousarray(np.rot90(x, 1))), -1)
    assert np.abs(p - q).max() < 1e-9 and np.abs(p - r).max() < 1e-9
    assert p.min() >= 0.0 and p.max() <= 1.0 + 1e-12
