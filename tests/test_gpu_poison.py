"""GPU: every scratch workspace poisoned (0xFF = NaN / -1) between calls -- a forward or a pass that reads a location it has
not written itself fails its bit comparison HERE, deterministically, instead of once in a while on recycled device memory.

Round 3 recorded one transient wrong result (an f32 prediction 3.8e-4 off on the first call of a fresh 1600-patch handle with n = 2,
DESIGN "Incidents"); repeat trials in fresh processes (zeroed VRAM) could never show a read of a never-written location.  The suspects
were the workspaces no call initialises: activation ping-pong buffers, the pooled-tile strips that live in them (row / column / corner
strips of sepconv_ws_kernel, finished by pool_fix_add_kernel), M-tail rows of the convolution tiles, patch_in / patch_out, the
thinning kernel's halo, the per-pass buffers when a pass holds fewer images than its capacity.  tmat_debug_poison fills all of them.
Reference layers: fl_tissue_model_tools/models.py:119-166."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CFG = dict(graph_thresh_1=5, graph_thresh_2=10, graph_smoothing_window=12, min_branch_length=12,
           remove_isolated_branches=False)


@pytest.fixture(scope="module")
def patches_and_ref(weights):
    from oracle import unet as ou
    rs = np.random.RandomState(21)
    x = rs.uniform(0, 1, (17, 320, 320)).astype(np.float32)
    x[5] = 0.0
    x[9, 40:300, 10:200] = 1.0
    return x, ou.forward_exact(weights, x)


@pytest.mark.parametrize("max_patches", [1600, 8])
def test_unet_forward_on_poisoned_workspaces(weights, patches_and_ref, max_patches):
    """n = 1, 2, 3, 17 on the bench-sized handle (1600 patches: the failing configuration of the incident, smallest n on the largest
    workspace) and on an 8-patch handle (n = 17 walks the patch list in chunks of 8, 8, 1); poison before EVERY call, twice over
    with two patterns (0xFF: NaN; 0x7F: large finite floats -- a max-pooling over a stale value would survive a NaN test only by luck)"""
    from tmat_amd import synth, _lib
    x, ref = patches_and_ref
    h = _lib.Handle(synth.pack_weights(weights), 0, max_patches)
    try:
        for pattern in (0xFF, 0x7F):
            for n in (1, 2, 3, 17):
                h.debug_poison(pattern)
                got = h.unet_predict(x[:n])
                assert not np.isnan(got).any(), f"pattern {pattern:#x} n {n}: NaN in the output"
                bad = np.flatnonzero((got.view(np.uint32) != ref[:n].view(np.uint32)).reshape(n, -1).any(axis=1))
                assert bad.size == 0, f"pattern {pattern:#x} n {n}: patches {bad} differ from the oracle"
    finally:
        h.close()


def test_analyze_pass_on_poisoned_workspaces(weights):
    """a pass with fewer images than its capacity, then a non-multiple-of-16 geometry, then the first geometry again, all on one
    bench-sized handle with every workspace (device and pinned) poisoned before each call: rows equal the oracle's"""
    from oracle import pipeline
    from tmat_amd import _lib, branches, synth
    imgs = np.stack([synth.synth_image(30 + i, 256, n_vessels=10, scale=1.0) for i in range(3)])       # 32 patches per image
    odd = synth.synth_image(40, 300, n_vessels=10, scale=1.0)[:250]                                    # 250 x 300 -> 188 x 156
    want = [pipeline.analyze_image(im, weights, CFG, 250.0) for im in imgs]
    want_odd = pipeline.analyze_image(odd, weights, CFG, 300.0)
    h = _lib.Handle(synth.pack_weights(weights), 0, 1600)
    try:
        first = branches.analyze_batch(h, imgs, CFG, 250.0)             # allocates the per-pass buffers (capacity: 3 images)
        for pattern in (0xFF, 0x7F):
            h.debug_poison(pattern)
            rows = branches.analyze_batch(h, imgs[:2], CFG, 250.0)      # fewer images than the pass capacity
            for r, w in zip(rows, want[:2]):
                assert (r[1], r[2], r[3]) == tuple(w)
            h.debug_poison(pattern)
            pred = np.empty((3, 160, 160), np.float64)
            _lib.check(_lib.lib().tmat_segment_batch(h.raw, _lib.ptr(imgs), 3, 256, 256, 0.625, _lib.ptr(pred)), "segment")
            assert not np.isnan(pred).any()
            h.debug_poison(pattern)
            rows = branches.analyze_batch(h, imgs, CFG, 250.0)
            for r, w in zip(rows, want):
                assert (r[1], r[2], r[3]) == tuple(w)
            assert [r[1:] for r in rows] == [r[1:] for r in first]
            r_odd = branches.analyze_batch(h, odd[None], CFG, 300.0)[0]            # new geometry: buffers are re-made ...
            h.debug_poison(pattern)                                                 # ... and poisoned
            r_odd2 = branches.analyze_batch(h, odd[None], CFG, 300.0)[0]
            assert (r_odd[1], r_odd[2], r_odd[3]) == tuple(want_odd) and r_odd2[1:] == r_odd[1:]
    finally:
        h.close()
