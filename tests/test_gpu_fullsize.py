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


def test_config3_chain_zproj_then_branches_on_device(weights):
    """BASELINE config #3 shape: a 2048 x 2048 Z stack is projected on the device (tmat_zproj_dev) and the projection is
    analysed without leaving HBM (tmat_analyze_batch_dev, 648 patches per image).  The device chain must equal the
    host-pointer chain (tmat_zproj_batch -> tmat_analyze_batch), and the projection must equal the oracle on crops."""
    import ctypes as C
    from oracle import zproj as oz
    from tmat_amd import _lib, branches, synth
    L = _lib.lib()
    h = _lib.Handle(synth.pack_weights(weights), 0, 648)
    try:
        base = synth.synth_image(3, 1024)
        big = np.kron(base, np.ones((2, 2), np.uint16))                   # 2048 x 2048
        rs = np.random.RandomState(0)
        stack = np.stack([big, (big // 2 + rs.randint(0, 200, big.shape)).astype(np.uint16), big[::-1].copy()])
        Z, H, W = stack.shape
        din, dproj = C.c_void_p(), C.c_void_p()
        _lib.check(L.tmat_dev_alloc(h.raw, stack.nbytes, C.byref(din)), "alloc")
        _lib.check(L.tmat_dev_alloc(h.raw, H * W * 2, C.byref(dproj)), "alloc")
        _lib.check(L.tmat_dev_upload(h.raw, din, stack.ctypes.data_as(C.c_void_p), stack.nbytes), "upload")
        _lib.check(L.tmat_zproj_dev(h.raw, din, 1, Z, H, W, 0, dproj), "zproj")
        row_dev = branches.analyze_batch(h, (1, H, W), CFG, 2000.0, dev_ptr=dproj.value)[0]
        proj = h.zproj(stack[None], "fs")
        row_host = branches.analyze_batch(h, proj, CFG, 2000.0)[0]
        assert row_dev == row_host and row_dev[1] > 0
        for (y, x) in ((0, 0), (900, 1100), (H - 128, W - 128)):
            want = oz.proj_focus_stacking(stack[:, y:y + 128, x:x + 128])
            ys = slice(0 if y == 0 else 4, 128 if y + 128 == H else 124)
            xs = slice(0 if x == 0 else 4, 128 if x + 128 == W else 124)
            assert np.array_equal(proj[0][y:y + 128, x:x + 128][ys, xs], want[ys, xs])
        _lib.check(L.tmat_dev_free(h.raw, din), "free")
        _lib.check(L.tmat_dev_free(h.raw, dproj), "free")
    finally:
        h.close()
