"""CPU checks of oracle/resnet.py (the invasion-depth classifier's restatement; reference fl_tissue_model_tools/models.py:33-82).
PARITY UNPINNED against Keras (TensorFlow absent, the reference holds no weights or fixtures for this model): what can be pinned here is
the restatement against itself -- the im2col layout of the 7x7 stride-2 stem against a direct gather, the stem's MFMA-order chain against a
float64 evaluation of ZeroPadding2D(3) + Conv2D(64, 7, strides 2) + folded BN + ReLU, the zero padding of pool and stem, and that a small
trunk runs end to end with probabilities strictly inside (0, 1)."""
import numpy as np

from oracle import resnet as orr


def _rand_input(seed, n=2, s=32):
    rs = np.random.RandomState(seed)
    return (rs.uniform(0, 255, (n, s, s, 1)) - np.array(orr.MEANS_BGR)).astype(np.float32)


def test_stem_im2col_layout():
    x = _rand_input(0)
    col = orr.stem_im2col(x)
    n, s = x.shape[0], x.shape[1]
    assert col.shape == (n, s // 2, s // 2, orr.STEM_K) and col.dtype == np.float32
    assert not col[..., orr.STEM_TAPS:].any()                      # the padding of K to six chunks of 32 is zeros
    rs = np.random.RandomState(1)
    for _ in range(200):
        i, yo, xo = rs.randint(n), rs.randint(s // 2), rs.randint(s // 2)
        ky, kx, c = rs.randint(7), rs.randint(7), rs.randint(3)
        iy, ix = 2 * yo + ky - 3, 2 * xo + kx - 3
        want = x[i, iy, ix, c] if 0 <= iy < s and 0 <= ix < s else np.float32(0)
        assert col[i, yo, xo, (ky * 7 + kx) * 3 + c] == want


def test_stem_against_float64():
    rs = np.random.RandomState(2)
    x = _rand_input(3)
    w = (rs.randn(7, 7, 3, 64) * 0.05).astype(np.float32)
    sc = (1 + 0.1 * rs.randn(64)).astype(np.float32)
    sh = (0.1 * rs.randn(64)).astype(np.float32)
    got = orr.stem(x, w, sc, sh)
    n, s = x.shape[0], x.shape[1]
    so = s // 2
    xp = np.zeros((n, s + 6, s + 6, 3), np.float64)
    xp[:, 3:-3, 3:-3] = x
    acc = np.zeros((n, so, so, 64))
    for ky in range(7):
        for kx in range(7):
            win = xp[:, ky:ky + 2 * so:2, kx:kx + 2 * so:2]
            for c in range(3):
                acc += win[..., c:c + 1] * w[ky, kx, c].astype(np.float64)
    want = np.maximum(acc * sc + sh, 0)
    assert got.shape == want.shape and got.dtype == np.float32
    assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max())
    assert (got >= 0).all() and (got > 0).any() and (got == 0).any()       # the ReLU is live on both sides


def test_pool_pads_with_zeros():
    x = -np.ones((1, 4, 4, 2), np.float32)                        # all negative: the zero padding wins at the border, not -inf
    p = orr.pool(x)
    assert p.shape == (1, 2, 2, 2)
    assert p[0, 0, 0, 0] == 0.0 and p[0, 1, 1, 0] == -1.0         # (0, 0) touches the padding, (1, 1) covers rows / columns 1..3 only


def test_small_trunk_runs():
    from tmat_amd import inv_depth
    w = inv_depth.synth_resnet_weights(5, "conv2_block1_out")
    x = _rand_input(6, n=2, s=32)
    p = orr.forward(w, x)
    assert p.shape == (2,) and p.dtype == np.float32
    assert np.all(p > 0) and np.all(p < 1)
    assert np.array_equal(p, orr.forward(w, x))                    # deterministic
