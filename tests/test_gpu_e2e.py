"""GPU parity, whole path through the C-ABI: uint16 images -> (count, total, average) rows and every
intermediate boundary (segment, postprocess) against the oracle, bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CFG = dict(graph_thresh_1=5, graph_thresh_2=10, graph_smoothing_window=12, min_branch_length=12,
           remove_isolated_branches=False)


@pytest.fixture(scope="module")
def images():
    from tmat_amd import synth
    return np.stack([synth.synth_image(i, 512, n_vessels=12, scale=1.0) for i in range(2)])


@pytest.fixture(scope="module")
def oracle_runs(weights, images):
    """oracle (exact-arithmetic UNet) results per image, computed once"""
    from oracle import pipeline
    return [pipeline.analyze_image(img, weights, CFG, 500.0, return_intermediates=True) for img in images]


def test_segment_batch_bitexact(handle, images, oracle_runs):
    from tmat_amd import _lib
    n, H, W = images.shape
    pred = np.empty((n, 320, 320), np.float64)
    _lib.check(_lib.lib().tmat_segment_batch(handle.raw, _lib.ptr(images), n, H, W, 0.625, _lib.ptr(pred)), "segment")
    for i in range(n):
        ref = oracle_runs[i][1]["pred"]
        assert np.array_equal(pred[i].view(np.uint64), ref.view(np.uint64)), f"image {i}"


def test_postprocess_batch_bitexact(handle, images, oracle_runs):
    from tmat_amd import _lib
    pred = np.stack([r[1]["pred"] for r in oracle_runs])
    field = np.empty((len(pred), 384, 384), np.float32)
    _lib.check(_lib.lib().tmat_postprocess_batch(handle.raw, _lib.ptr(pred), len(pred), 320, 320, 384, 384, _lib.ptr(field)), "post")
    for i in range(len(pred)):
        assert np.array_equal(field[i].view(np.uint32), oracle_runs[i][1]["field"].view(np.uint32))


def test_analyze_batch_rows_match_oracle(handle, images, oracle_runs):
    from tmat_amd import branches
    rows = branches.analyze_batch(handle, images, CFG, 500.0, first_index=7)
    assert sum(r[1] for r in rows) > 0, "test images should produce branches"
    for i, (idx, cnt, tot, avg) in enumerate(rows):
        n0, tot0, avg0 = oracle_runs[i][0]
        assert idx == 7 + i
        assert cnt == n0, f"image {i}: count {cnt} != {n0}"
        assert tot == tot0 and avg == avg0, f"image {i}: {tot} {avg} vs {tot0} {avg0}"


def test_analyze_nonsquare_and_degenerate(handle, weights):
    from oracle import pipeline
    from tmat_amd import branches, synth
    img = synth.synth_image(11, 320, n_vessels=12)[:256]          # (256, 320)
    flat = np.full((256, 320), 1000, np.uint16)                    # constant image
    zero = np.zeros((256, 320), np.uint16)
    batch = np.stack([img, flat, zero])
    rows = branches.analyze_batch(handle, batch, CFG, 300.0)
    for i in range(3):
        n0, tot0, avg0 = pipeline.analyze_image(batch[i], weights, CFG, 300.0)
        assert rows[i][1] == n0 and rows[i][2] == tot0 and rows[i][3] == avg0


def test_mirror_api(handle, weights, images, oracle_runs):
    """the reference-named Python surface drives the same kernels"""
    from oracle import pipeline, morph
    from tmat_amd import models, smooth_tiled_predictions as stp, synth
    seg = models.UNetXceptionPatchSegmentor(320, synth.pack_weights(weights), [64, 128, 256, 512], ds_ratio=0.625,
                                            max_patches=96)
    small = morph.lanczos4_resize_u16(images[0], (320, 320))
    x = morph.rescale_intensity(small, (0, 1)).astype(np.float32)
    p1 = seg.predict(x, auto_resample=False)
    p2 = stp.predict_img_with_smooth_windowing(x, 320, 2, seg.model.predict)
    ref = oracle_runs[0][1]["pred"]
    assert np.array_equal(p1, ref) and np.array_equal(p2, ref)
    y = seg.model.predict(np.zeros((2, 320, 320), np.float32))
    assert y.shape == (2, 320, 320, 1)
    with pytest.raises(TypeError):
        stp.predict_img_with_smooth_windowing(x, 320, 2, lambda b, verbose=0: b)
    seg.handle.close()


RCCL_CHILD = r"""
import ctypes as C, sys
sys.path[:0] = [r"{repo}", r"{repo}/tissue-model-analysis-tools_amd"]
from tmat_amd import _lib
L = _lib.lib()
handle = _lib.Handle(None, 0)
try:
    rccl = C.CDLL("librccl.so.1")
except OSError:
    rccl = C.CDLL("/opt/rocm/lib/librccl.so")
hip = C.CDLL("libamdhip64.so")

class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]
uid, comm = UniqueId(), C.c_void_p()
assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
rows = (_lib.Row * 5)()
for i in range(5):
    rows[i].index, rows[i].count, rows[i].total_px, rows[i].avg_px = 100 + i, 3 * i, 1.5 * i, 0.25 * i
nbytes = C.sizeof(rows)
din, dout = C.c_void_p(), C.c_void_p()
_lib.check(L.tmat_dev_alloc(handle.raw, nbytes, C.byref(din)), "alloc")
_lib.check(L.tmat_dev_alloc(handle.raw, nbytes, C.byref(dout)), "alloc")
_lib.check(L.tmat_dev_upload(handle.raw, din, C.byref(rows), nbytes), "upload")
_lib.check(L.tmat_gather_rows(comm, din, 5, dout, None), "gather")
assert hip.hipDeviceSynchronize() == 0
got = (_lib.Row * 5)()
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
assert hip.hipMemcpy(C.byref(got), dout, nbytes, 2) == 0           # hipMemcpyDeviceToHost
assert [(r.index, r.count, r.total_px, r.avg_px) for r in got] == [(r.index, r.count, r.total_px, r.avg_px) for r in rows]
rccl.ncclCommDestroy.argtypes = [C.c_void_p]
rccl.ncclCommDestroy(comm)
_lib.check(L.tmat_dev_free(handle.raw, din), "free")
_lib.check(L.tmat_dev_free(handle.raw, dout), "free")
handle.close()
print("gathered 5 rows over RCCL")
"""


def test_gather_rows_over_rccl_single_rank(tmp_path):
    """tmat_gather_rows on a one-rank RCCL communicator created through ctypes: the gathered rows are the local rows.  Runs as a
    host program of its own (a child process with the ROCm installation's librccl and nothing else): a process that has
    already imported torch has torch's bundled ROCm runtime mapped, and a system librccl loaded after it cannot initialise."""
    import subprocess
    import sys
    from pathlib import Path
    script = tmp_path / "rccl_child.py"
    script.write_text(RCCL_CHILD.replace("{repo}", str(Path(__file__).resolve().parents[1])))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "gathered 5 rows over RCCL" in r.stdout, r.stdout + r.stderr


STREAM_CHILD = r"""
import json, sys
sys.path[:0] = [r"{repo}", r"{repo}/tissue-model-analysis-tools_amd"]
import numpy as np
from tmat_amd import _lib, branches, synth
h = _lib.Handle(synth.pack_weights(synth.synth_weights(0)), 0, 144)          # 72 patches per 512 x 512 image: two images per pass, five passes
imgs = np.stack([synth.synth_image(100 + i, 512, n_vessels=12, scale=1.0) for i in range(9)])
cfg = dict(graph_thresh_1=5, graph_thresh_2=10, graph_smoothing_window=12, min_branch_length=12, remove_isolated_branches=False)
print(json.dumps([[int(r[0]), int(r[1]), repr(float(r[2])), repr(float(r[3]))] for r in branches.analyze_batch(h, imgs, cfg, 500.0)]))
h.close()
"""


def test_stream_layouts_give_identical_rows(tmp_path):
    """the pipeline's stream layouts are scheduling only: the tail of a pass on the second stream (default), everything on the main
    stream (TMAT_TAIL_STREAM=0) and the two-stream network (TMAT_STREAMS=2) return the same rows over five passes of 9 images"""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    script = tmp_path / "streams_child.py"
    script.write_text(STREAM_CHILD.replace("{repo}", str(Path(__file__).resolve().parents[1])))
    outs = []
    for extra in ({}, {"TMAT_TAIL_STREAM": "0"}, {"TMAT_STREAMS": "2"}):
        r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900, env=dict(os.environ, **extra))
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1] == outs[2] and len(outs[0]) == 9 and sum(r[1] for r in outs[0]) > 0


def test_rows_within_north_star_tolerance_of_the_as_written_network(handle, weights, images):
    """The north-star bar against the reference CPU path: integer branch counts equal, branch lengths within 1e-4
    relative.  The closest thing to that path available here is the oracle with the as-written network (PyTorch-CPU
    convolutions in library order: 9-tap convolutions over the upsampled tensors, residual 1x1 after the upsampling);
    the GPU path (sub-pixel form, hoisted residual, fixed FMA order) must agree with it to that bar, and its
    probability map must stay within 2e-5 of it."""
    from oracle import pipeline
    from tmat_amd import _lib, branches
    rows = branches.analyze_batch(handle, images, CFG, 500.0)
    pred = np.empty((len(images), 320, 320), np.float64)
    _lib.check(_lib.lib().tmat_segment_batch(handle.raw, _lib.ptr(np.ascontiguousarray(images)), len(images), 512, 512, 0.625,
                                             _lib.ptr(pred)), "segment")
    for i, img in enumerate(images):
        (n0, tot0, avg0), inter = pipeline.analyze_image(img, weights, CFG, 500.0, unet_kind="torch", return_intermediates=True)
        assert np.abs(pred[i] - inter["pred"]).max() < 2e-5
        assert rows[i][1] == n0
        assert rows[i][2] == pytest.approx(tot0, rel=1e-4) and rows[i][3] == pytest.approx(avg0, rel=1e-4)


def test_images_that_need_more_patches_than_the_workspace(weights, images, oracle_runs):
    """max_patches only sizes the activation workspace: an image that needs more patches (72 here, workspace 40) runs
    its patch list in chunks, one image per pass, with the same rows / probability maps as the oracle"""
    from tmat_amd import _lib, branches, synth
    h = _lib.Handle(synth.pack_weights(weights), 0, 40)
    try:
        rows = branches.analyze_batch(h, images, CFG, 500.0)
        for i, r in enumerate(rows):
            assert (r[1], r[2], r[3]) == tuple(oracle_runs[i][0]), i
        pred = np.empty((len(images), 320, 320), np.float64)
        _lib.check(_lib.lib().tmat_segment_batch(h.raw, _lib.ptr(np.ascontiguousarray(images)), len(images), 512, 512, 0.625,
                                                 _lib.ptr(pred)), "segment")
        for i in range(len(images)):
            assert np.array_equal(pred[i].view(np.uint64), oracle_runs[i][1]["pred"].view(np.uint64))
        # five oversize passes in a row: every pass's whole network runs in the front half and writes the single patch_out that the
        # previous pass's blend reads on the second stream -- the front half waits for that blend (pipeline.cpp:enqueue_front)
        five = np.ascontiguousarray(images[[0, 1, 1, 0, 1]])
        rows5 = branches.analyze_batch(h, five, CFG, 500.0)
        for r, i in zip(rows5, [0, 1, 1, 0, 1]):
            assert (r[1], r[2], r[3]) == tuple(oracle_runs[i][0]), i
        # a second geometry on the same handle afterwards (buffers regrow / are reused)
        small = synth.synth_image(4, 256, n_vessels=10, scale=1.0)
        big = _lib.Handle(synth.pack_weights(weights), 0, 128)
        try:
            assert branches.analyze_batch(h, small[None], CFG, 250.0) == branches.analyze_batch(big, small[None], CFG, 250.0)
        finally:
            big.close()
    finally:
        h.close()


def test_eight_bit_sources_take_the_fixed_point_path(handle, weights):
    """cv2.resize runs uint8 images through its fixed-point Lanczos path (11-bit coefficients, int32 accumulation,
    (v + 2^21) >> 22, saturation at 255) -- tmat_set_input_depth / input_bits=8; the float path is for uint16 only."""
    from oracle import morph, pipeline
    from tmat_amd import _lib, branches, synth
    img16 = synth.synth_image(1, 512, n_vessels=12, scale=1.0)
    img8 = np.clip(img16.astype(np.float64) / 120.0, 0, 255).astype(np.uint8)      # large saturated areas
    assert (img8 == 255).mean() > 0.01
    wide = img8.astype(np.uint16)
    s8 = morph.lanczos4_resize_u8(wide, (320, 320))
    s16 = morph.lanczos4_resize_u16(wide, (320, 320))
    assert s16.max() > 255 and s8.max() == 255                                   # the overshoot exists and is clipped
    assert (s8.astype(int) != np.minimum(s16, 255).astype(int)).any()             # and the two arithmetic paths do differ
    # the resized image itself, through the segment entry point: compare the probability maps bit for bit
    L = _lib.lib()
    pred = np.empty((1, 320, 320), np.float64)
    _lib.check(L.tmat_set_input_depth(handle.raw, 8), "depth")
    _lib.check(L.tmat_segment_batch(handle.raw, _lib.ptr(wide[None]), 1, 512, 512, 0.625, _lib.ptr(pred)), "segment")
    _lib.check(L.tmat_set_input_depth(handle.raw, 16), "depth")
    want = pipeline.segment(wide, weights, input_bits=8)
    assert np.array_equal(pred[0].view(np.uint64), want.view(np.uint64))
    row8 = branches.analyze_batch(handle, wide[None], CFG, 500.0, input_bits=8)[0]
    assert row8[1:] == tuple(pipeline.analyze_image(wide, weights, CFG, 500.0, input_bits=8))
    row16 = branches.analyze_batch(handle, wide[None], CFG, 500.0)[0]             # and the default depth is restored per call
    assert row16[1:] == tuple(pipeline.analyze_image(wide, weights, CFG, 500.0))


def test_nonsquare_segment_follows_cv2_dsize_order(handle, weights):
    """compute_branches.py:309-312 passes (round(H r), round(W r)) to cv2.resize as dsize = (width, height): a 256 x 320
    image becomes 200 rows x 160 columns before the network sees it"""
    from oracle import morph, pipeline
    from tmat_amd import _lib, synth
    img = np.ascontiguousarray(synth.synth_image(11, 320, n_vessels=12)[:256])          # (256, 320)
    assert morph.target_shape(img.shape, 0.625) == (160, 200) and morph.resized_shape(img.shape, 0.625) == (200, 160)
    pred = np.empty((1, 200, 160), np.float64)
    _lib.check(_lib.lib().tmat_segment_batch(handle.raw, _lib.ptr(img[None]), 1, 256, 320, 0.625, _lib.ptr(pred)), "segment")
    want = pipeline.segment(img, weights)
    assert want.shape == (200, 160) and np.array_equal(pred[0].view(np.uint64), want.view(np.uint64))


def test_malformed_weight_blobs_are_refused(weights):
    """TMATW001 parsing: offsets / counts that would wrap, truncated tables and tensors whose shapes do not fit the layer
    plan are refused with a message instead of being read out of bounds"""
    import struct
    from tmat_amd import _lib, synth
    blob = bytearray(synth.pack_weights(weights))

    def refuse(b, what):
        with pytest.raises(Exception, match=what):
            _lib.Handle(bytes(b), 0, 8).close()
    refuse(blob[:90], "truncated")
    refuse(blob[:100], "bad entry|truncated")
    bad = bytearray(blob); struct.pack_into("<Q", bad, 16 + 76, 0xFFFFFFFFFFFFFFF0)        # count of entry 0: off + cnt * 4 wraps
    refuse(bad, "bad entry")
    bad = bytearray(blob); struct.pack_into("<Q", bad, 16 + 68, 0xFFFFFFFFFFFFFFFC)        # offset of entry 0 beyond the blob
    refuse(bad, "bad entry")
    w2 = dict(weights); w2["down0.bn1"] = w2["down0.bn1"][:, :64].copy()                  # BN block of the wrong width
    refuse(synth.pack_weights(w2), "does not match|unexpected shape")
    w3 = dict(weights); del w3["up2.res.b"]
    refuse(synth.pack_weights(w3), "missing tensor")
