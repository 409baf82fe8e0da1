"""GPU: the drop-in scripts/compute_branches.py end to end (argument surface, utf-16 CSV with the reference's
header, -N suffixing, config.json, threshold grid files, error exits)."""
import csv
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parents[1]
SCRIPT = REPO / "tissue-model-analysis-tools_amd" / "scripts" / "compute_branches.py"


def run(args, env_extra=None):
    env = dict(os.environ, TMAT_SYNTHETIC_WEIGHTS="1")
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(SCRIPT)] + args, capture_output=True, text=True, env=env, timeout=600)


def read_csv(path):
    with open(path, encoding="utf-16") as f:
        return list(csv.reader(f))


def test_script_end_to_end(tmp_path, handle):
    from tmat_amd import branches, synth
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    imgs = {f"img_{i}": synth.synth_image(i, 512, n_vessels=12, scale=1.0) for i in range(3)}
    for k, v in imgs.items():
        np.save(ind / f"{k}.npy", v)
    r = run([str(ind), str(outd), "--image-width-microns", "500"])
    assert r.returncode == 0, r.stdout + r.stderr
    rows = read_csv(outd / "branching_analysis.csv")
    assert rows[0] == ["Image", "Total # of branches", "Total branch length (µm)", "Average branch length (µm)"]
    cfg = dict(graph_thresh_1=5, graph_thresh_2=10, graph_smoothing_window=12, min_branch_length=12)
    want = branches.analyze_batch(handle, np.stack([imgs[k] for k in sorted(imgs)]), cfg, 500.0)
    assert [r_[0] for r_ in rows[1:]] == sorted(imgs)
    for got, w in zip(rows[1:], want):
        assert int(got[1]) == w[1]
        assert float(got[2]) == pytest.approx(branches.pixels_to_microns(w[2], 384, 500.0), rel=1e-12)
        assert float(got[3]) == pytest.approx(branches.pixels_to_microns(w[3], 384, 500.0), rel=1e-12)
    saved = json.loads((outd / "config.json").read_text())
    assert saved["graph_thresh_1"] == 5 and saved["image_width_microns"] == 500.0 and "max_branch_length" not in saved
    # second run into the same directory: nothing is overwritten (reference helper.get_unique_output_filepath semantics)
    r = run([str(ind), str(outd), "--image-width-microns", "500", "--graph-thresh-1", "2", "5"])
    assert r.returncode == 0, r.stdout + r.stderr
    assert (outd / "config-2.json").is_file()
    assert (outd / "branching_analysis_CONFIG_thresh1_2.0.csv").is_file() and (outd / "branching_analysis_CONFIG_thresh1_5.0.csv").is_file()
    assert read_csv(outd / "branching_analysis_CONFIG_thresh1_5.0.csv")[1:] == rows[1:]


def test_script_two_ranks_on_one_gpu_write_the_one_process_csv(tmp_path):
    """the script's N > 1 flow with the real library (TMAT_DIST_REHEARSE=1: both ranks on device 0, the row gather over gloo because RCCL
    refuses two ranks on one device): 5 images -- an uneven split -- sharded, analysed, gathered, written by rank 0; the CSV equals the
    one-process CSV byte for byte"""
    import socket
    from tmat_amd import synth
    ind = tmp_path / "in"
    ind.mkdir()
    for i in range(5):
        np.save(ind / f"img_{i}.npy", synth.synth_image(60 + i, 512, n_vessels=12, scale=1.0))
    r = run([str(ind), str(tmp_path / "w1"), "--image-width-microns", "500"])
    assert r.returncode == 0, r.stdout + r.stderr
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ, TMAT_SYNTHETIC_WEIGHTS="1", TMAT_DIST_REHEARSE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(SCRIPT), str(ind), str(tmp_path / "w2"), "--image-width-microns", "500"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    a = (tmp_path / "w1" / "branching_analysis.csv").read_bytes()
    b = (tmp_path / "w2" / "branching_analysis.csv").read_bytes()
    assert a == b and len(read_csv(tmp_path / "w2" / "branching_analysis.csv")) == 6


def test_script_error_exits(tmp_path):
    r = run([str(tmp_path / "missing"), str(tmp_path / "o")])
    assert r.returncode == 1 and "does not exist" in r.stdout
    (tmp_path / "in").mkdir()
    np.save(tmp_path / "in" / "a.npy", np.zeros((64, 64), np.uint16))
    r = run([str(tmp_path / "in"), str(tmp_path / "o")])                       # no --image-width-microns
    assert r.returncode == 1 and "image-width-microns" in r.stdout
    r = run([str(tmp_path / "in"), str(tmp_path / "o"), "-c", str(tmp_path / "nope.json")])
    assert r.returncode == 1 and "Config file" in r.stdout


def test_script_takes_the_width_from_tiff_metadata(tmp_path, handle):
    """no --image-width-microns: width = columns x PhysicalPixelSizes.X of each file (compute_branches.py:184-212);
    files with different pixel sizes are analysed with their own pixel parameters"""
    from PIL import Image, TiffImagePlugin
    from tmat_amd import branches, synth
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    imgs = {f"m_{i}": synth.synth_image(10 + i, 512, n_vessels=12, scale=1.0) for i in range(2)}
    px_um = {"m_0": 0.9765625, "m_1": 1.5}               # 512 px -> 500 um and 768 um
    for k, v in imgs.items():
        info = TiffImagePlugin.ImageFileDirectory_v2()
        info[270] = ('<OME><Image><Pixels DimensionOrder="XYZCT" SizeX="512" SizeY="512" '
                     f'PhysicalSizeX="{px_um[k]}" PhysicalSizeY="{px_um[k]}" Type="uint16"/></Image></OME>')
        Image.fromarray(v).save(ind / f"{k}.tif", tiffinfo=info)
    r = run([str(ind), str(outd)])
    assert r.returncode == 0, r.stdout + r.stderr
    rows = read_csv(outd / "branching_analysis.csv")[1:]
    cfg = dict(graph_thresh_1=5, graph_thresh_2=10, graph_smoothing_window=12, min_branch_length=12)
    for got, k in zip(rows, sorted(imgs)):
        width = 512 * px_um[k]
        w = branches.analyze_batch(handle, imgs[k][None], cfg, width)[0]
        assert got[0] == k and int(got[1]) == w[1]
        assert float(got[2]) == pytest.approx(branches.pixels_to_microns(w[2], 384, width), rel=1e-12)


def _oracle_stack_row(stack, width_um, hessian="gaussian_derivatives", t1=5, t2=10):
    from oracle import sato as osato
    from tmat_amd import branches
    cfg = {"graph_thresh_1": t1, "graph_thresh_2": t2, "graph_smoothing_window": 12, "min_branch_length": 12, "remove_isolated_branches": False}
    n, tot, avg = osato.analyze_stack(stack, cfg, width_um, hessian=hessian)
    return n, branches.pixels_to_microns(tot, 384, width_um), branches.pixels_to_microns(avg, 384, width_um)


def test_script_analyzes_z_stacks_through_the_sato_branch(tmp_path):
    """a directory of z-numbered slice sequences (compute_branches.py:547-553) goes through the Z-stack branch: one CSV row per
    stack, equal to the oracle's restatement of the branch; --visualizations writes the branch's two pictures"""
    from PIL import Image
    from tmat_amd import synth
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    stacks = {"wellA": synth.synth_stack(1, 4, 200, 256, n_vessels=8), "wellB": synth.synth_stack(2, 3, 200, 256, n_vessels=8)}
    for k, st in stacks.items():
        for z, sl in enumerate(st):
            Image.fromarray(sl).save(ind / f"{k}_z{z}.tif")
    r = run([str(ind), str(outd), "--image-width-microns", "800", "--visualizations"])
    assert r.returncode == 0, r.stdout + r.stderr
    rows = read_csv(outd / "branching_analysis.csv")
    assert [x[0] for x in rows[1:]] == ["wellA", "wellB"]
    for got in rows[1:]:
        n, tot, avg = _oracle_stack_row(stacks[got[0]], 800.0)
        assert int(got[1]) == n and float(got[2]) == pytest.approx(tot, rel=1e-12) and float(got[3]) == pytest.approx(avg, rel=1e-12)
        assert n > 0
    for k, st in stacks.items():
        vdir = outd / "visualizations" / k
        assert sorted(p.name for p in vdir.iterdir()) == ["original_image.png", "vesselness_image.png"]
        mx = st.max(0).astype(np.float64)
        want = np.rint((mx - mx.min()) / (mx.max() - mx.min()) * 255.0).astype(np.uint8)
        assert np.array_equal(np.array(Image.open(vdir / "original_image.png")), want)
        assert np.array(Image.open(vdir / "vesselness_image.png")).shape == (300, 384)


def test_script_detect_well_on_a_z_stack(tmp_path):
    """-w for Z-stack inputs (compute_branches.py:227-243): the CSV row equals the oracle's row with the pruning mask, well_mask.png is
    the oracle's well mask"""
    from PIL import Image
    from oracle import sato as osato
    from test_gpu_sato import _well_stack
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    st = _well_stack(11)
    for z, sl in enumerate(st):
        Image.fromarray(sl).save(ind / f"plate_z{z}.tif")
    r = run([str(ind), str(outd), "--image-width-microns", "1000", "-w", "--well-seed", "3", "--visualizations"])
    assert r.returncode == 0, r.stdout + r.stderr
    rows = read_csv(outd / "branching_analysis.csv")
    cfg = dict(graph_thresh_1=5, graph_thresh_2=10, graph_smoothing_window=12, min_branch_length=12, remove_isolated_branches=False)
    (n, tot, avg), well, _ = osato.analyze_stack(st, cfg, 1000.0, detect_well=True, well_seed=3, return_masks=True)
    from tmat_amd import branches
    got = rows[1]
    assert got[0] == "plate" and int(got[1]) == n and n > 0
    assert float(got[2]) == pytest.approx(branches.pixels_to_microns(tot, 384, 1000.0), rel=1e-12)
    assert np.array_equal(np.array(Image.open(outd / "visualizations" / "plate" / "well_mask.png")) > 0, well)


def test_script_multipage_stacks_threshold_grid_and_legacy_hessian(tmp_path):
    """multi-page TIFF files are stacks too (compute_branches.py:554-559); a threshold grid writes one CSV per configuration
    from ONE vesselness image per stack; --sato-hessian gradient selects the scikit-image <= 0.19 Hessian"""
    from PIL import Image
    from tmat_amd import synth
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    st = synth.synth_stack(5, 5, 180, 240, n_vessels=8)
    pages = [Image.fromarray(sl) for sl in st]
    pages[0].save(ind / "stackfile.tif", save_all=True, append_images=pages[1:])
    r = run([str(ind), str(outd), "--image-width-microns", "600", "--graph-thresh-1", "2", "5", "--sato-hessian", "gradient"])
    assert r.returncode == 0, r.stdout + r.stderr
    for t1 in (2, 5):
        rows = read_csv(outd / f"branching_analysis_CONFIG_thresh1_{t1}.0.csv")
        n, tot, avg = _oracle_stack_row(st, 600.0, "gradient", t1=t1)
        assert rows[1][0] == "stackfile" and int(rows[1][1]) == n and float(rows[1][2]) == pytest.approx(tot, rel=1e-12)


def test_script_visualizations_and_time_refusal(tmp_path, handle, weights):
    """--visualizations writes the reference's four image dumps per image (compute_branches.py:315, 331, 347, 348); the
    prediction picture is the 0..255 rescale of the GPU path's own probability map.  --time other than 0 is refused."""
    from PIL import Image
    from oracle import pipeline
    from tmat_amd import synth
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    img = synth.synth_image(3, 512, n_vessels=12, scale=1.0)
    np.save(ind / "v_0.npy", img)
    r = run([str(ind), str(outd), "--image-width-microns", "500", "--visualizations"])
    assert r.returncode == 0, r.stdout + r.stderr
    vdir = outd / "visualizations" / "v_0"
    names = sorted(p.name for p in vdir.iterdir())
    assert names == ["distance_transform.png", "original_image.png", "prediction.png", "segmentation_mask.png"]
    pred = pipeline.segment(img, weights)
    want = np.rint((pred - pred.min()) / (pred.max() - pred.min()) * 255.0).astype(np.uint8)
    assert np.array_equal(np.array(Image.open(vdir / "prediction.png")), want)
    seg = np.array(Image.open(vdir / "segmentation_mask.png"))
    assert set(np.unique(seg)) <= {0, 255} and seg.shape == pred.shape
    r = run([str(ind), str(tmp_path / "o2"), "--image-width-microns", "500", "--time", "2"])
    assert r.returncode == 1 and "Time 2 is out of range" in r.stdout          # the reference's message (helper.py:62-66)


def test_script_time_series_tiff_and_detect_well(tmp_path, handle, weights):
    """--time N picks the frame of a time-series TIFF (helper.load_image :57-84); -w / --detect-well runs the well-mask form
    of the 2-D branch (compute_branches.py:318-337) with an explicit --well-seed: rows equal the staged API's"""
    from PIL import Image, TiffImagePlugin
    from tmat_amd import branches, synth
    sys.path.insert(0, str(REPO / "tests"))
    from test_gpu_wellmask import _well_image
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    frames = [synth.synth_image(40, 512, n_vessels=12, scale=1.0), _well_image(5)]
    ifd = TiffImagePlugin.ImageFileDirectory_v2()
    ifd[270] = "ImageJ=1.53\nimages=2\nframes=2\nhyperstack=true\n"
    Image.fromarray(frames[0]).save(ind / "series.tif", save_all=True, append_images=[Image.fromarray(frames[1])], tiffinfo=ifd)
    r = run([str(ind), str(outd), "--image-width-microns", "500"])
    assert r.returncode == 1 and "time series image but no time index" in r.stdout
    cfg = dict(graph_thresh_1=5, graph_thresh_2=10, graph_smoothing_window=12, min_branch_length=12)
    for t, extra in ((0, []), (1, ["-w", "--well-seed", "7"])):
        r = run([str(ind), str(outd / f"t{t}"), "--image-width-microns", "500", "--time", str(t)] + extra)
        assert r.returncode == 0, r.stdout + r.stderr
        rows = read_csv(outd / f"t{t}" / "branching_analysis.csv")
        if extra:
            want = branches.well_rows(handle, branches.well_fields(handle, frames[t][None], 0.625, 16, 7, warn=lambda m: None), cfg, 500.0)[0]
        else:
            want = branches.analyze_batch(handle, frames[t][None], cfg, 500.0)[0]
        assert rows[1][0] == "series" and int(rows[1][1]) == want[1] and (want[1] > 0 or extra)
        assert float(rows[1][2]) == pytest.approx(branches.pixels_to_microns(want[2], 384, 500.0), rel=1e-12)
