"""CPU: the Z-projection oracle (oracle/zproj.py) against an independent scipy.ndimage evaluation of the same
kernels, its defining properties, and the host-side stack discovery of tmat_amd/zstacks.py.

OpenCV is absent here and the reference holds no fixture for compute_zproj, so these tests pin the restatement to the
published kernels (parity unpinned against cv2 itself; see oracle/zproj.py)."""
import json
import os
import sys
from pathlib import Path

import numpy as np
import pytest
from scipy import ndimage as ndi

from oracle import zproj as oz

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO / "tissue-model-analysis-tools_amd"))


@pytest.mark.parametrize("shape", [(37, 50), (8, 5), (3, 2), (1, 9), (64, 64)])
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_blur_and_laplacian_match_scipy_mirror_correlation(shape, dtype):
    rs = np.random.RandomState(shape[0] * 100 + shape[1])
    img = rs.randint(0, np.iinfo(dtype).max + 1, shape).astype(dtype)
    g = np.outer(oz.G5, oz.G5)
    want_blur = ((ndi.correlate(img.astype(np.int64), g, mode="mirror") + 128) >> 8).astype(dtype)
    got_blur = oz.gaussian_blur5(img)
    assert np.array_equal(got_blur, want_blur)
    want_lap = ndi.correlate(got_blur.astype(np.int64), oz.K5, mode="mirror").astype(np.float64)
    assert np.array_equal(oz.laplacian5(got_blur), want_lap)
    # the Laplacian kernel is d2/dx2 + d2/dy2 of the aperture-5 Sobel family and sums to zero
    assert oz.K5.sum() == 0 and np.array_equal(oz.K5, oz.K5.T)
    assert np.abs(oz.K5).sum() * 65535 < 2 ** 24          # exact in float32, as the module header claims


def test_focus_stacking_properties():
    rs = np.random.RandomState(0)
    st = rs.randint(0, 65536, (5, 40, 33)).astype(np.uint16)
    out = oz.proj_focus_stacking(st)
    assert out.dtype == np.uint16 and out.shape == (40, 33)
    assert np.all((st == out[None]).any(axis=0))                       # every pixel comes from its own column of the stack
    assert np.array_equal(oz.proj_focus_stacking(st[:1]), st[0])       # one slice: identity
    same = np.repeat(st[:1], 4, axis=0)
    assert np.array_equal(oz.proj_focus_stacking(same), st[0])         # ties keep the first slice
    flat = np.stack([np.full((9, 9), v, np.uint16) for v in (7, 3, 9)])
    assert np.array_equal(oz.proj_focus_stacking(flat), flat[0])       # zero focus everywhere: strict '>' never fires again
    # a sharp slice among blurred copies wins where it has detail
    base = rs.randint(0, 60000, (64, 64)).astype(np.uint16)
    soft = ndi.uniform_filter(base.astype(np.float64), 7).astype(np.uint16)
    out = oz.proj_focus_stacking(np.stack([soft, base, soft]))
    assert (out == base).mean() > 0.8
    assert np.array_equal(oz.proj_focus_stacking(np.moveaxis(st, 0, 2), axis=2), oz.proj_focus_stacking(st))


GOLD = json.loads((Path(__file__).parent / "golden" / "zstacks.json").read_text())


@pytest.mark.parametrize("name", sorted(GOLD["layouts"]))
def test_discovery_matches_reference_goldens(tmp_path, name):
    """tests/golden/zstacks.json was produced by the reference's own zstacks.py (tools/make_goldens.py zstacks)"""
    from tmat_amd import zstacks as zs
    rec = GOLD["layouts"][name]
    for f in rec["files"]:
        (tmp_path / f).parent.mkdir(parents=True, exist_ok=True)
        (tmp_path / f).write_bytes(b"x")
    if "error" in rec:
        with pytest.raises(zs.ZStackInputException):
            zs.find_zstack_image_sequences(str(tmp_path))
    else:
        got = zs.find_zstack_image_sequences(str(tmp_path))
        assert {k: [os.path.relpath(p, tmp_path).replace(os.sep, "/") for p in v] for k, v in got.items()} == rec["sequences"]
    if "files_as_stacks" in rec:
        assert {k: os.path.relpath(v, tmp_path) for k, v in zs.find_zstack_files(str(tmp_path)).items()} == rec["files_as_stacks"]


def test_clean_ids_and_reductions_match_reference_goldens():
    from tmat_amd import zstacks as zs
    for case in GOLD["clean_ids"]:
        assert zs.clean_zstack_ids(list(case["in"])) == case["out"], case
    for rec in GOLD["proj"].values():
        st = np.array(rec["stack"], np.uint16)
        for m in ("min", "max", "avg", "med"):
            got = getattr(oz, "proj_" + m)(st)
            assert str(got.dtype) == rec[m]["dtype"] and np.array_equal(got, np.array(rec[m]["values"])), m


def test_discovery_of_image_sequences_and_files(tmp_path):
    from tmat_amd import zstacks as zs
    d = tmp_path / "seq"
    d.mkdir()
    for well in ("A1", "B2"):
        for z in (10, 2, 1):
            (d / f"{well}_z{z}_ch0.tif").write_bytes(b"x")
    got = zs.find_zstack_image_sequences(str(d))
    assert sorted(got) == ["A1_ch0", "B2_ch0"]
    assert [os.path.basename(p) for p in got["A1_ch0"]] == ["A1_z1_ch0.tif", "A1_z2_ch0.tif", "A1_z10_ch0.tif"]
    # one folder per stack
    d2 = tmp_path / "nested"
    for well in ("w1", "w2"):
        (d2 / well).mkdir(parents=True)
        for z in range(3):
            (d2 / well / f"img_Z{z:02d}.png").write_bytes(b"x")
    got = zs.find_zstack_image_sequences(str(d2))
    assert len(got) == 2 and all(len(v) == 3 for v in got.values())
    # duplicated slice numbers are refused
    d3 = tmp_path / "dup"
    d3.mkdir()
    (d3 / "a_z1.tif").write_bytes(b"x")
    (d3 / "a_Z1.tif").write_bytes(b"x")
    with pytest.raises(zs.ZStackInputException):
        zs.find_zstack_image_sequences(str(d3))
    # multi-page files: one stack per file, id = stem
    d4 = tmp_path / "files"
    d4.mkdir()
    (d4 / "s1.tif").write_bytes(b"x")
    (d4 / "s2.tiff").write_bytes(b"x")
    assert sorted(zs.find_zstack_files(str(d4))) == ["s1", "s2"]


def test_zproj_script_argument_surface():
    sys.path.insert(0, str(REPO / "tissue-model-analysis-tools_amd" / "scripts"))
    import compute_zproj as cz
    a = cz.parse_zproj_args(["'in dir'", "out", "-m", "fs", "--channel", "1"])
    assert (a.in_root, a.out_root, a.method, a.channel, a.time, a.area) == ("in dir", "out", "fs", 1, None, False)
    assert cz.parse_zproj_args(["i", "o"]).method == "max"
    with pytest.raises(SystemExit):
        cz.parse_zproj_args(["i", "o", "-m", "sum"])
