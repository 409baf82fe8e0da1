"""Opt-in split-precision MFMA path (tmat_set_precision("bf16x3"), include/tmat.h): not bit-exact by construction, gated by
the tolerance BASELINE.json's north_star states for the whole path -- integer branch counts equal, branch lengths within
1e-4 relative -- against the f32 path (which is bit-exact with the oracle) and against the as-written PyTorch-CPU graph.
The default (f32) path is untouched: the last test checks that switching back restores the bit-exact results."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CFG = dict(graph_thresh_1=5, graph_thresh_2=10, graph_smoothing_window=12, min_branch_length=12,
           remove_isolated_branches=False)
LEN_RTOL = 1e-4        # north_star: "branch-length floats within 1e-4 relative"
PRED_ATOL = 2e-4       # probability maps: what the split-precision products may move a sigmoid output by (observed ~1e-5)


@pytest.fixture()
def alt_handle(weights):
    from tmat_amd import synth, _lib
    h = _lib.Handle(synth.pack_weights(weights), 0, 1600)
    yield h
    h.close()


def test_patches_close_to_f32_and_to_the_as_written_graph(alt_handle, weights):
    from oracle import unet as ou
    rs = np.random.RandomState(21)
    x = rs.uniform(0, 1, (4, 320, 320)).astype(np.float32)
    x[1, :, 100:] = 0.0
    f32 = alt_handle.unet_predict(x)
    alt_handle.set_precision("bf16x3")
    alt = alt_handle.unet_predict(x)
    d = float(np.abs(alt.astype(np.float64) - f32).max())
    print(f"\nbf16x3 vs f32 on 4 patches: max |d pred| = {d:.3e}, differing outputs {int((alt != f32).sum())} of {alt.size}")
    assert d < PRED_ATOL
    assert (alt != f32).any(), "bf16x3 returned the f32 bits: the split-precision kernels did not run"
    ref = ou.forward_torch(weights, x)
    d_t = float(np.abs(alt.astype(np.float64) - ref).max())
    print(f"bf16x3 vs the as-written PyTorch-CPU graph: max |d pred| = {d_t:.3e}")
    assert d_t < PRED_ATOL


def test_random_weights_every_channel_live(weights):
    """O(1) random weights in every tensor: an operand mapping error in the bf16 fragments (wrong k order between A and B,
    hi / lo planes swapped) would show as O(1) errors here, not as 1e-5"""
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
        f32 = h.unet_predict(x)
        h.set_precision("bf16x3")
        alt = h.unet_predict(x)
    finally:
        h.close()
    d = float(np.abs(alt.astype(np.float64) - f32).max())
    print(f"\nrandom weights: max |d pred| = {d:.3e}")
    assert d < 1e-3


def test_rows_within_north_star_tolerance_on_bench_images(alt_handle):
    """16 images of bench.py's workload (1024 x 1024, SURVEY 8d generator): counts equal, total / average length within 1e-4
    relative of the f32 path"""
    from tmat_amd import branches, synth
    imgs = np.stack([synth.synth_image(i, 1024) for i in range(16)])
    rows32 = branches.analyze_batch(alt_handle, imgs, CFG, 1000.0)
    alt_handle.set_precision("bf16x3")
    rows16 = branches.analyze_batch(alt_handle, imgs, CFG, 1000.0)
    bad_counts = [(a[0], a[1], b[1]) for a, b in zip(rows32, rows16) if a[1] != b[1]]
    rel = [abs(b[2] - a[2]) / max(abs(a[2]), 1e-30) for a, b in zip(rows32, rows16) if a[1] == b[1] and a[1] > 0]
    print(f"\nbf16x3 vs f32 on 16 bench images: count mismatches {len(bad_counts)} {bad_counts}, max relative length difference {max(rel) if rel else 0:.3e}")
    assert sum(r[1] for r in rows32) > 0
    assert not bad_counts, f"branch counts differ: {bad_counts}"
    assert max(rel) <= LEN_RTOL


def test_switching_back_restores_the_bit_exact_path(alt_handle, weights):
    from oracle import unet as ou
    x = np.random.RandomState(2).uniform(0, 1, (2, 320, 320)).astype(np.float32)
    alt_handle.set_precision("bf16x3")
    alt_handle.unet_predict(x)
    alt_handle.set_precision("f32")
    got = alt_handle.unet_predict(x)
    ref = ou.forward_exact(weights, x)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    with pytest.raises(ValueError):
        alt_handle.set_precision("fp8")
