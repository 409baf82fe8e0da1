"""GPU parity: binary morphology kernels (threshold, median, labelling, perimeter, Zhang thinning, fork test,
EDT) through tmat_filter_edt_batch against oracle/morph.py and the reference-generated goldens, bit-exact."""
from pathlib import Path

import numpy as np
import pytest
from scipy import ndimage as ndi

pytestmark = pytest.mark.gpu

G = np.load(Path(__file__).parent / "golden" / "filter.npz")


def unpack(name, key):
    shape = tuple(G[name + "_shape"])
    return np.unpackbits(G[name + "_" + key])[: shape[0] * shape[1]].reshape(shape).astype(bool)


def as_pred(mask, rs):
    """a probability map whose > 0.5 set is exactly `mask`"""
    return np.where(mask, rs.uniform(0.5000001, 1.0, mask.shape), rs.uniform(0.0, 0.5, mask.shape))


@pytest.mark.parametrize("name", ["d5", "m1", "blobs", "noise", "small", "empty", "full"])
def test_filter_and_edt_match_reference_goldens(handle, name):
    from oracle import morph
    rs = np.random.RandomState(1)
    m = unpack(name, "mask")
    pred = as_pred(m, rs)[None]
    filt, dist = handle.filter_edt(pred)
    assert np.array_equal(filt[0], unpack(name, "filtered"))          # the reference's own filter_branch_seg_mask
    ref = ndi.distance_transform_edt(filt[0])
    assert np.array_equal(dist[0].view(np.uint64), ref.view(np.uint64))
    osk, odist = morph.medial_axis(filt[0])
    assert np.array_equal(dist[0], odist)


def test_batch_of_mixed_images(handle):
    """several images per launch: per-image flags (thinning convergence, EDT 'no background' quirk) stay separate"""
    from oracle import morph
    rs = np.random.RandomState(2)
    masks = [ndi.gaussian_filter(rs.normal(size=(96, 128)), s) > t for s, t in ((3, 0.02), (1.5, 0.0), (6, -0.01))]
    masks += [np.ones((96, 128), bool), np.zeros((96, 128), bool)]
    pred = np.stack([as_pred(m, rs) for m in masks])
    filt, dist = handle.filter_edt(pred)
    for i, m in enumerate(masks):
        want = morph.filter_branch_seg_mask(pred[i] > 0.5)
        assert np.array_equal(filt[i], want), i
        assert np.array_equal(dist[i], ndi.distance_transform_edt(want)), i


@pytest.mark.parametrize("shape,out", [((96, 128), (72, 96)), ((128, 128), (128, 128)), ((200, 150), (96, 72)), ((64, 64), (100, 100))])
def test_finish_stage_matches_oracle(handle, shape, out):
    """a15-a17 on the GPU (EDT of the inverted skeleton, centre-line weighting, gaussian + linear zoom + clip,
    rescale to 0..255) through tmat_finish_batch, bit for bit against oracle/morph.py"""
    from oracle import morph
    rs = np.random.RandomState(5)
    masks = [ndi.gaussian_filter(rs.normal(size=shape), s) > t for s, t in ((3, 0.02), (5, 0.0))]
    masks += [np.zeros(shape, bool), np.ones(shape, bool)]
    pred = np.stack([as_pred(m, rs) for m in masks])
    filt = [morph.filter_branch_seg_mask(p > 0.5) for p in pred]
    sk_dist = [morph.medial_axis(f) for f in filt]
    skel = np.stack([s for s, _ in sk_dist])
    dist = np.stack([d for _, d in sk_dist])
    field, f255 = handle.finish(pred, dist, skel, out)
    for i in range(len(masks)):
        want, _, wskel = morph.postprocess(pred[i], out)
        assert np.array_equal(wskel, skel[i])
        assert np.array_equal(field[i].view(np.uint32), want.view(np.uint32)), i
        w255 = morph.rescale_intensity(want, (0, 255)).astype(np.float32)
        assert np.array_equal(f255[i].view(np.uint32), np.ascontiguousarray(w255).view(np.uint32)), i


@pytest.mark.parametrize("use_median,remove_isolated", [(True, True), (False, True), (True, False), (False, False)])
def test_filter_branch_seg_mask_mirror_options(use_median, remove_isolated):
    """tmat_amd.transforms.filter_branch_seg_mask (the reference seam transforms.py:306) runs on the GPU for every
    combination of its options (compute_branches.py:293 uses footprint=None, remove_isolated=False), single mask and batch"""
    import sys
    from pathlib import Path as P
    sys.path.insert(0, str(P(__file__).resolve().parents[1] / "tissue-model-analysis-tools_amd"))
    from oracle import morph
    from tmat_amd import transforms
    rs = np.random.RandomState(11)
    masks = [ndi.gaussian_filter(rs.normal(size=(90, 120)), s) > t for s, t in ((3, 0.02), (1.2, 0.05), (5, -0.01))]
    masks += [unpack("d5", "mask")[:90, :120], np.zeros((90, 120), bool)]
    fp = transforms.disk(2) if use_median else None
    got = transforms.filter_branch_seg_mask(np.stack(masks), fp, remove_isolated)
    for i, m in enumerate(masks):
        want = morph.filter_branch_seg_mask(m, use_median, remove_isolated)
        assert np.array_equal(got[i], want), i
    assert np.array_equal(transforms.filter_branch_seg_mask(masks[0], fp, remove_isolated), got[0])


@pytest.mark.parametrize("name", ["d5", "m1", "blobs", "noise", "small", "empty", "full"])
def test_device_medial_axis_matches_reference_goldens(handle, name):
    """tmat_medial_axis_batch (EDT + key build + radix sort + ordered thinning, all on the device; only the RandomState(0)
    permutation comes from the host) against skimage 0.18.3's medial_axis run on the reference's filtered masks
    (tools/make_goldens.py filter): skeleton exact, distance map exact."""
    import hashlib
    filt = unpack(name, "filtered")
    skel, dist = handle.medial_axis(filt[None])
    assert np.array_equal(skel[0], unpack(name, "ma_skel"))
    sha = np.frombuffer(hashlib.sha256(np.ascontiguousarray(dist[0]).tobytes()).digest(), np.uint8)
    assert np.array_equal(sha, G[name + "_ma_dist_sha"])


def test_device_medial_axis_batch_and_adversarial_order(handle):
    """a batch of different images per launch, including masks where almost every pixel of the visiting order touches
    its predecessor (thin diagonal lines, a checker of 2 x 2 blocks): the claim / cut logic of the thinning kernel then
    advances a few lanes at a time and must still reproduce the sequential result"""
    from oracle import morph
    rs = np.random.RandomState(5)
    H, W = 120, 150
    yy, xx = np.mgrid[:H, :W]
    masks = [ndi.gaussian_filter(rs.normal(size=(H, W)), 3) > 0.0,
             (np.abs(yy - xx) % 7 < 2),                          # bundles of 2-px diagonals
             ((yy // 2 + xx // 2) % 2 == 0),                     # 2 x 2 checker: every pixel has equal distance 1
             np.ones((H, W), bool),
             np.zeros((H, W), bool),
             rs.uniform(size=(H, W)) > 0.3]
    skel, dist = handle.medial_axis(np.stack(masks))
    for i, m in enumerate(masks):
        osk, odist = morph.medial_axis(m)
        assert np.array_equal(dist[i], odist), i
        assert np.array_equal(skel[i], osk), i


def test_device_medial_axis_deep_blobs_repeatable(handle):
    """large blobs at the pipeline's size: the dependency DAG is hundreds of levels deep, so the wavefront needs many
    launches and tile halos matter (a workgroup must never finish a pixel whose predecessor its neighbour cannot finish
    in the same launch).  Three runs of a batch of 8, each compared with the host implementation."""
    from tmat_amd import _lib
    rs = np.random.RandomState(1)
    masks = np.stack([ndi.gaussian_filter(rs.normal(size=(640, 640)), s) > t
                      for s, t in ((8, 0.0), (5, 0.01), (12, -0.005), (3, 0.02), (20, 0.0), (6, 0.0), (9, 0.01), (4, -0.01))])
    ref = [_lib.host_medial_axis(m)[0] for m in masks]
    for rep in range(3):
        sk, _ = handle.medial_axis(masks)
        for i in range(len(masks)):
            assert np.array_equal(sk[i], ref[i]), (rep, i, int((sk[i] != ref[i]).sum()))
