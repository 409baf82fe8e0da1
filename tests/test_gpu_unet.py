"""GPU parity: HIP UNet + smooth tiled prediction vs the oracle, bit-exact (through the C-ABI)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_unet_patch_bitexact(handle, weights):
    from oracle import unet as ou
    rs = np.random.RandomState(3)
    x = rs.uniform(0, 1, (3, 320, 320)).astype(np.float32)
    x[1] = 0.0                      # all-zero patch
    x[2, :160] = 1.0                # saturated half
    ref = ou.forward_exact(weights, x)
    got = handle.unet_predict(x)
    assert got.shape == ref.shape
    nbad = int((got.view(np.uint32) != ref.view(np.uint32)).sum())
    assert nbad == 0, f"{nbad} of {got.size} outputs differ, max |d| = {np.abs(got - ref).max()}"


def test_unet_random_weights_bitexact():
    """O(1) random weights in every tensor (not the structured-synthetic set): exercises every
    channel of every MFMA tile."""
    from oracle import unet as ou
    from tmat_amd import synth, _lib
    rs = np.random.RandomState(5)
    w = synth.synth_weights(1)
    for k in w:
        if k.rsplit(".", 1)[-1].startswith("bn"):
            C = w[k].shape[1]
            w[k][0] = rs.uniform(0.5, 1.5, C); w[k][1] = rs.normal(0, 0.3, C)
            w[k][2] = rs.normal(0, 0.3, C); w[k][3] = rs.uniform(0.5, 1.5, C)
        else:
            fan = int(np.prod(w[k].shape[:-1])) if w[k].ndim > 1 else 1
            w[k] = rs.normal(0, 1.0 / np.sqrt(max(fan, 1)), w[k].shape).astype(np.float32)
    h = _lib.Handle(synth.pack_weights(w), 0, 16)
    try:
        x = rs.uniform(0, 1, (2, 320, 320)).astype(np.float32)
        ref = ou.forward_exact(w, x)
        got = h.unet_predict(x)
    finally:
        h.close()
    nbad = int((got.view(np.uint32) != ref.view(np.uint32)).sum())
    assert nbad == 0, f"{nbad} of {got.size} outputs differ, max |d| = {np.abs(got - ref).max()}"


def test_predict_smooth_bitexact(handle, weights):
    from oracle import unet as ou, blend
    rs = np.random.RandomState(4)
    x = rs.uniform(0, 1, (200, 180)).astype(np.float32)
    ref = blend.predict_img_with_smooth_windowing(x, 320, 2, ou.predict_exact(weights))
    got = handle.predict_smooth(x)
    assert got.dtype == np.float64 and got.shape == ref.shape
    nbad = int((got.view(np.uint64) != ref.view(np.uint64)).sum())
    assert nbad == 0, f"{nbad} of {got.size} differ, max |d| = {np.abs(got - ref).max()}"


def test_large_launch_race_screen(weights):
    """A launch that fills the chip several times over (1024 patches: 5000 workgroups per conv launch and more), made of
    copies of 4 patches laid out so that copies land in different tiles / workgroups / XCDs.  Every copy must equal
    the oracle bit for bit, three times in a row: an LDS-DMA ordering bug shows up as rare wrong tiles, not as a crash."""
    from oracle import unet as ou
    from tmat_amd import synth, _lib
    rs = np.random.RandomState(9)
    base = rs.uniform(0, 1, (4, 320, 320)).astype(np.float32)
    ref = ou.forward_exact(weights, base)
    n = 1024
    idx = (np.arange(n) * 7 + (np.arange(n) // 13)) % 4
    x = base[idx]
    h = _lib.Handle(synth.pack_weights(weights), 0, n)
    try:
        for rep in range(3):
            got = h.unet_predict(x)
            bad = np.flatnonzero((got.view(np.uint32) != ref[idx].view(np.uint32)).reshape(n, -1).any(axis=1))
            assert bad.size == 0, f"rep {rep}: {bad.size} of {n} patches differ (first {bad[:8]})"
    finally:
        h.close()


@pytest.mark.parametrize("env", [{"TMAT_FUSED_POOL": "0"}, {"TMAT_FUSED_SEP": "0"}, {"TMAT_FUSED_SEP": "0", "TMAT_RELU_COPY": "0"},
                                 {"TMAT_STEM_FUSED": "0"}, {"TMAT_RELU_COPY": "0"}, {"TMAT_STEM_FUSED": "0", "TMAT_RELU_COPY": "0", "TMAT_FUSED_POOL": "0"}])
def test_unfused_down_path_variants_give_the_same_bits(weights, env, monkeypatch):
    """the network has several forms, selected per handle at creation: separate depthwise / pointwise / pool kernels (TMAT_FUSED_SEP=0,
    the tested reference form) or the fused depthwise -> pointwise kernel (default) with or without the max-pool + residual add behind it
    (TMAT_FUSED_POOL=0); the stem recomputed inside the first separable convolution (round 4, default) or written by its own kernel
    (TMAT_STEM_FUSED=0); activated copies of the block outputs (default) or ReLU on load (TMAT_RELU_COPY=0); all must equal the oracle"""
    from oracle import unet as ou
    from tmat_amd import synth, _lib
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rs = np.random.RandomState(11)
    x = rs.uniform(0, 1, (3, 320, 320)).astype(np.float32)
    x[2, 100:220, 100:220] = 0.0
    ref = ou.forward_exact(weights, x)
    h = _lib.Handle(synth.pack_weights(weights), 0, 8)
    try:
        got = h.unet_predict(x)
    finally:
        h.close()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
