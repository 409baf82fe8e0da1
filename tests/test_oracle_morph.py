"""CPU: oracle/morph.py against golden vectors produced by the reference's transforms.py running on
scikit-image 0.18.3 (tools/make_goldens.py filter)."""
import hashlib

import numpy as np
import pytest

from oracle import morph

G = np.load(__import__("pathlib").Path(__file__).parent / "golden" / "filter.npz")
NAMES = ["d5", "m1", "blobs", "noise", "small", "empty", "full"]


def unpack(name, key):
    shape = tuple(G[name + "_shape"])
    return np.unpackbits(G[name + "_" + key])[: shape[0] * shape[1]].reshape(shape).astype(bool)


@pytest.mark.parametrize("name", NAMES)
def test_median_label_perimeter_skeleton(name):
    m = unpack(name, "mask")
    med = morph.median13(m)
    assert np.array_equal(med, unpack(name, "median"))
    assert np.array_equal(morph.skeletonize_zhang(med), unpack(name, "skel"))
    from scipy import ndimage as ndi
    lab, n = ndi.label(med, np.ones((3, 3)))
    assert n == int(G[name + "_nlabels"])
    areas = np.array([int((lab[sl] == i).sum()) for i, sl in enumerate(ndi.find_objects(lab), 1)], np.int64)
    perims = np.array([morph.perimeter4(lab[sl] == i) for i, sl in enumerate(ndi.find_objects(lab), 1)])
    assert np.array_equal(areas, G[name + "_areas"])
    np.testing.assert_allclose(perims, G[name + "_perims"], rtol=1e-13, atol=0)


@pytest.mark.parametrize("name", NAMES)
def test_filter_branch_seg_mask(name):
    med = unpack(name, "median")
    assert np.array_equal(morph.filter_branch_seg_mask(med, use_median=False), unpack(name, "filtered"))
    assert np.array_equal(morph.filter_branch_seg_mask(med, use_median=False, remove_isolated=False),
                          unpack(name, "filtered_keepiso"))
    assert np.array_equal(morph.filter_branch_seg_mask(unpack(name, "mask")), unpack(name, "filtered"))


@pytest.mark.parametrize("name", NAMES)
def test_medial_axis(name):
    filt = unpack(name, "filtered")
    sk, dist = morph.medial_axis(filt)
    assert np.array_equal(sk, unpack(name, "ma_skel"))
    sha = np.frombuffer(hashlib.sha256(np.ascontiguousarray(dist).tobytes()).digest(), np.uint8)
    assert np.array_equal(sha, G[name + "_ma_dist_sha"])
