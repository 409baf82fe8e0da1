"""GPU parity: Z projections (csrc/zproj_kernels.hip) through tmat_zproj_batch against oracle/zproj.py, bit-exact,
including image borders, odd sizes, ties, uint8 stacks, the full config-#3 stack size (through crops the oracle can
check in seconds and through size-independent properties), and the drop-in script end to end."""
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def plain():
    sys.path.insert(0, str(REPO / "tissue-model-analysis-tools_amd"))
    from tmat_amd import _lib
    h = _lib.Handle(None, 0)
    yield h
    h.close()


@pytest.mark.parametrize("shape", [(5, 37, 50), (3, 64, 64), (4, 3, 2), (2, 1, 9), (7, 33, 129), (1, 40, 40)])
@pytest.mark.parametrize("dtype", [np.uint16, np.uint8])
def test_focus_stacking_matches_oracle(plain, shape, dtype):
    from oracle import zproj as oz
    rs = np.random.RandomState(sum(shape))
    stacks = rs.randint(0, np.iinfo(dtype).max + 1, (3,) + shape).astype(dtype)
    stacks[1, :, : shape[1] // 2] = stacks[1, :1, : shape[1] // 2]      # identical slices in a region: ties keep slice 0
    stacks[2] = 1234 % (np.iinfo(dtype).max + 1)                        # flat stack: zero focus everywhere
    got = plain.zproj(stacks, "fs")
    assert got.dtype == dtype and got.shape == (3,) + shape[1:]
    for i in range(3):
        assert np.array_equal(got[i], oz.proj_focus_stacking(stacks[i])), i


@pytest.mark.parametrize("method", ["min", "max", "avg", "med"])
@pytest.mark.parametrize("Z", [1, 2, 5, 16, 64, 65, 130])
def test_reductions_match_numpy(plain, method, Z):
    from oracle import zproj as oz
    rs = np.random.RandomState(Z)
    stacks = rs.randint(0, 65536, (2, Z, 45, 70)).astype(np.uint16)
    stacks[0, :, :8] = rs.randint(0, 3, (Z, 8, 70))          # heavy ties (the deep-stack median is a radix select: Z > 64)
    got = plain.zproj(stacks, method)
    want = np.stack([getattr(oz, "proj_" + method)(s) for s in stacks])
    assert got.dtype == want.dtype
    assert np.array_equal(got, want)


def test_model_entry_points_refuse_a_plain_handle(plain):
    with pytest.raises(Exception, match="no model"):
        plain.unet_predict(np.zeros((1, 320, 320), np.float32))


def test_full_size_stack_crops_and_properties(plain):
    """config #3 size: 16 slices of 2048 x 2048.  The oracle checks crops (corners keep the true image border, the
    interior crops are compared away from their own artificial border); the whole image is checked by properties."""
    from oracle import zproj as oz
    rs = np.random.RandomState(7)
    Z, H, W = 16, 2048, 2048
    st = rs.randint(0, 65536, (Z, H, W)).astype(np.uint16)
    st[5, 512:1024] = st[4, 512:1024]                    # two equal slices in a band
    got = plain.zproj(st[None], "fs")[0]
    assert np.all((st == got[None]).any(axis=0))         # every pixel is one of its own stack's values
    C = 160
    for (y, x) in ((0, 0), (0, W - C), (H - C, 0), (H - C, W - C), (700, 900), (1500, 40)):
        crop = st[:, y:y + C, x:x + C]
        want = oz.proj_focus_stacking(crop)
        ys = slice(0 if y == 0 else 4, C if y + C == H else C - 4)
        xs = slice(0 if x == 0 else 4, C if x + C == W else C - 4)
        assert np.array_equal(got[y:y + C, x:x + C][ys, xs], want[ys, xs]), (y, x)
    assert np.array_equal(plain.zproj(st[None, :1], "fs")[0], st[0])          # single slice: identity
    assert np.array_equal(plain.zproj(st[None], "max")[0], st.max(axis=0))


def test_script_end_to_end(tmp_path, plain):
    from PIL import Image
    from oracle import zproj as oz
    sys.path.insert(0, str(REPO / "tissue-model-analysis-tools_amd" / "scripts"))
    import compute_zproj as cz
    rs = np.random.RandomState(3)
    in_root, out_root = tmp_path / "in", tmp_path / "out"
    in_root.mkdir()
    stacks = {}
    for well in ("A1", "B7"):
        st = rs.randint(0, 65536, (4, 96, 80)).astype(np.uint16)
        stacks[well] = st
        for z in range(4):
            Image.fromarray(st[z]).save(in_root / f"{well}_z{z}.tif")
    cz.main(cz.parse_zproj_args([str(in_root), str(out_root), "-m", "fs"]))
    for well, st in stacks.items():
        got = np.array(Image.open(out_root / f"{well}_fs.tif"))
        assert got.dtype == np.uint16 and np.array_equal(got, oz.proj_focus_stacking(st))
    cz.main(cz.parse_zproj_args([str(in_root), str(out_root), "-m", "fs"]))          # second run: unique names
    assert (out_root / "A1_fs-2.tif").is_file()
    cz.main(cz.parse_zproj_args([str(in_root), str(out_root)]))                       # default method: max
    assert np.array_equal(np.array(Image.open(out_root / "B7_max.tif")), stacks["B7"].max(axis=0))


def test_script_time_series_and_area(tmp_path, plain):
    """--time N selects the T plane of an ImageJ hyperstack (reference helper.load_image), and -a/--area runs the cell-area drop-in on
    the projections with OUT_ROOT as its input and output directory (compute_zproj.py:98-119)"""
    import csv
    import subprocess
    from PIL import Image, TiffImagePlugin
    from oracle import cellarea as ca
    from tmat_amd import synth
    script = REPO / "tissue-model-analysis-tools_amd" / "scripts" / "compute_zproj.py"
    in_root, out_root = tmp_path / "in", tmp_path / "out"
    in_root.mkdir()
    T, Z = 2, 3
    vol = np.stack([np.stack([synth.synth_image(50 + 10 * t + z, 256, n_vessels=10, scale=0.5) for z in range(Z)]) for t in range(T)])      # (T, Z, H, W)
    ifd = TiffImagePlugin.ImageFileDirectory_v2()
    ifd[270] = f"ImageJ=1.53\nimages={T * Z}\nslices={Z}\nframes={T}\nhyperstack=true\n"
    pages = [Image.fromarray(vol[t, z]) for t in range(T) for z in range(Z)]          # ImageJ order: Z runs faster than T
    pages[0].save(in_root / "plate.tif", save_all=True, append_images=pages[1:], tiffinfo=ifd)
    r = subprocess.run([sys.executable, str(script), str(in_root), str(out_root)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 1 and "time series image but no time index" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([sys.executable, str(script), str(in_root), str(out_root), "--time", "1", "-m", "max", "-a"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    proj = np.array(Image.open(out_root / "plate_max.tif"))
    assert np.array_equal(proj, vol[1].max(axis=0))
    rows = list(csv.reader(open(out_root / "calculations" / "cell_area.csv")))
    assert rows[0] == ["image_id", "area_pct"] and rows[1][0] == "plate_max"
    oa, ok = ca.cell_area(proj, 512, 0.0)
    assert float(rows[1][1]) == oa * 100
    assert np.array_equal(np.array(Image.open(out_root / "thresholded" / "plate_max_thresholded.png")), ok)
