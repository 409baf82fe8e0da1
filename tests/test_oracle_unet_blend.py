"""CPU: oracle/blend.py against goldens from the imported reference smooth_tiled_predictions
(bit-exact), and the exact-arithmetic UNet oracle against the independent PyTorch restatement."""
import hashlib
from pathlib import Path

import numpy as np
import pytest

from make_goldens import pred_toy
from oracle import blend, unet as ou
from tmat_amd import synth

G = np.load(Path(__file__).parent / "golden" / "blend.npz")


def test_spline_window_bitexact():
    assert np.array_equal(blend.spline_window(320), G["window320"])
    assert np.array_equal(blend.spline_window(64), G["window64"])
    w2 = blend.window_2d(320)
    # the four half-overlapping windows sum to subdivisions**2 = 4
    s = w2[:160, :160] + w2[160:, :160] + w2[:160, 160:] + w2[160:, 160:]
    np.testing.assert_allclose(s, 4.0, rtol=0, atol=1e-12)


@pytest.mark.parametrize("name", ["toy_a", "toy_b", "toy_c"])
def test_smooth_windowing_bitexact(name):
    r = blend.predict_img_with_smooth_windowing(G[name + "_in"], int(G[name + "_ws"]), 2, pred_toy)
    assert np.array_equal(r, G[name + "_out"])


def test_identity_predictor_640():
    img = np.random.RandomState(11).uniform(0, 1, (640, 640)).astype(np.float32)
    r = blend.predict_img_with_smooth_windowing(img, 320, 2, lambda b, verbose=0: np.asarray(b)[..., None])
    sha = np.frombuffer(hashlib.sha256(np.ascontiguousarray(r).tobytes()).digest(), np.uint8)
    assert np.array_equal(sha, G["ident640_sha256"])
    assert np.abs(r - img).max() <= 2.3e-16


def test_d4_roundtrip():
    a = np.random.RandomState(0).uniform(size=(12, 17))
    assert np.array_equal(blend.d4_undo_mean(blend.d4_do(a)), (a * 8) / 8.0) or np.allclose(blend.d4_undo_mean(blend.d4_do(a)), a, atol=1e-15)


def _dense_random_weights(filters, seed):
    rs = np.random.RandomState(seed)
    w = synth.synth_weights(seed, filters)
    for k in w:
        if k.rsplit(".", 1)[-1].startswith("bn"):
            C = w[k].shape[1]
            w[k][0] = rs.uniform(0.5, 1.5, C); w[k][1] = rs.normal(0, 0.3, C)
            w[k][2] = rs.normal(0, 0.3, C); w[k][3] = rs.uniform(0.5, 1.5, C)
        else:
            fan = int(np.prod(w[k].shape[:-1])) if w[k].ndim > 1 else 1
            w[k] = rs.normal(0, 1.0 / np.sqrt(max(fan, 1)), w[k].shape).astype(np.float32)
    return w


def test_exact_unet_matches_torch_restatement_small():
    w = _dense_random_weights((8, 16, 32, 64), 1)
    x = np.random.RandomState(2).uniform(0, 1, (3, 32, 32)).astype(np.float32)
    a = ou.forward_exact(w, x)
    b = ou.forward_torch(w, x)
    assert a.shape == (3, 32, 32) and a.dtype == np.float32
    np.testing.assert_allclose(a, b, rtol=0, atol=2e-5)


def test_subpixel_form_equals_as_written_taps():
    """The 4-tap sub-pixel form of the first transposed convolution of up blocks 2.. is the as-written 9-tap
    convolution over the upsampled tensor up to f32 rounding of the pre-summed taps (and exactly equal when the
    taps are small integers, where no sum rounds)."""
    w = _dense_random_weights((8, 16, 32, 64), 4)
    x = np.random.RandomState(5).uniform(0, 1, (2, 32, 32)).astype(np.float32)
    ta, tb = {}, {}
    a = ou.forward_exact(w, x, taps=ta, subpixel=True)
    b = ou.forward_exact(w, x, taps=tb, subpixel=False)
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-5)
    for k in ta:
        np.testing.assert_allclose(ta[k], tb[k], rtol=0, atol=2e-5 * max(1.0, float(np.abs(tb[k]).max())))
    # exact case: integer-valued taps and inputs (every product and sum is exact in f32)
    rs = np.random.RandomState(6)
    W9 = rs.randint(-3, 4, (3, 3, 32, 64)).astype(np.float32)
    S = rs.randint(-4, 5, (2, 5, 7, 32)).astype(np.float32)
    one, zero = np.ones(64, np.float32), np.zeros(64, np.float32)
    full = ou._conv(S, W9, 3, 1, 1, 1, one, zero, None, 0, 0)
    sub = ou._conv_subpixel(S, W9, 1, one, zero, 0)
    assert np.array_equal(full, sub)


def test_exact_unet_full_size_one_patch():
    w = synth.synth_weights(0)
    x = np.random.RandomState(3).uniform(0, 1, (1, 320, 320)).astype(np.float32)
    a = ou.forward_exact(w, x)
    b = ou.forward_torch(w, x)
    np.testing.assert_allclose(a, b, rtol=0, atol=2e-5)
    assert 0.0 < a.min() and a.max() <= 1.0


def test_sigmoid_det_accuracy():
    import ctypes
    L = ou.lib()
    z = np.linspace(-30, 30, 2001)
    got = np.array([L.orc_sigmoid(ctypes.c_float(v)) for v in z])
    np.testing.assert_allclose(got, 1 / (1 + np.exp(-z)), rtol=1e-6, atol=1e-12)
