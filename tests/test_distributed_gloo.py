"""CPU, world_size 2 over gloo: the N>1 path returns the same rows as a single process.

* gather_rows: shard -> per-rank rows -> ONE fixed-size all-gather, on every rank, for even / ragged / empty shards.
* run_sharded: the driver scripts/compute_branches.py runs (shard the ids, load in bounded chunks, analyse per shape
  group, gather, rank 0 writes the CSV), with a stand-in for the GPU analyser (deterministic rows computed from the
  pixels): the CSV written under world 2 must equal the CSV written by one process."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]

WORKER = r"""
import os, sys, json
sys.path.insert(0, r"{repo}"); sys.path.insert(0, r"{repo}/tissue-model-analysis-tools_amd")
import torch.distributed as dist
from tmat_amd import distributed
dist.init_process_group("gloo")
ws, rank, _ = distributed.world()
n = int(sys.argv[1])
calls = []
orig = dist.all_gather_into_tensor
dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
mine = distributed.shard_indices(n, rank, ws)
rows = [(int(i), int(i) * 3 % 17, float(i) * 1.25, float(i) / 7.0) for i in mine]      # stand-in for analyze_batch rows
allrows = distributed.gather_rows(rows, n_total=n)
expect = [(i, i * 3 % 17, i * 1.25, i / 7.0) for i in range(n)]
assert allrows == expect, (rank, allrows[:3], expect[:3])
assert len(calls) == 1, calls                   # a single collective
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok", len(allrows))
"""

DRIVER = r"""
import csv, os, sys
sys.path.insert(0, r"{repo}"); sys.path.insert(0, r"{repo}/tissue-model-analysis-tools_amd")
import numpy as np
from tmat_amd import branches, distributed
ws, rank, _ = distributed.world()
if ws > 1:
    import torch.distributed as dist
    dist.init_process_group("gloo")
out = sys.argv[1]
ids = ["img%02d" % i for i in range(11)]
def load_fn(i):
    k = int(i[3:])
    rs = np.random.RandomState(k)
    shape = (48, 64) if k % 3 else (32, 32)              # two shape groups
    return rs.randint(0, 60000, shape).astype(np.uint16 if k % 4 else np.uint8)
def width_fn(i, img):
    return 500.0 if int(i[3:]) % 2 else 1000.0
def analyze_fn(batch, width_um, thresh, input_bits):        # stand-in for the GPU path: rows are a function of the pixels
    return [(k, int(im.sum() % 97), float(im.mean()) * thresh[0], float(im.std()) + input_bits) for k, im in enumerate(batch)]
cfg = dict(graph_thresh_1=[5, 7], graph_thresh_2=10)
res = branches.run_sharded(ids, load_fn, width_fn, analyze_fn, cfg, rank, ws, chunk=3, log=lambda m: None)
if rank == 0:
    for suffix, rows in res.items():
        with open(os.path.join(out, "rows" + suffix + ".csv"), "w") as f:
            w = csv.writer(f)
            for gidx, cnt, tot, avg in rows:
                w.writerow([ids[gidx], cnt, repr(tot), repr(avg)])
if ws > 1:
    dist.barrier(); dist.destroy_process_group()
print("rank", rank, "done")
"""


def _run(script, args, nproc):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if nproc == 1:
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            env.pop(k, None)
        cmd = [sys.executable, str(script)] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), str(script)] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


@pytest.mark.parametrize("n", [0, 5, 8])
def test_gather_rows_world2(tmp_path, n):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(repo=str(REPO)))
    assert _run(script, [str(n)], 2).count("ok") == 2


def test_run_sharded_world2_equals_world1(tmp_path):
    script = tmp_path / "driver.py"
    script.write_text(DRIVER.format(repo=str(REPO)))
    o1, o2 = tmp_path / "w1", tmp_path / "w2"
    o1.mkdir(); o2.mkdir()
    _run(script, [str(o1)], 1)
    assert _run(script, [str(o2)], 2).count("done") == 2
    files = sorted(p.name for p in o1.iterdir())
    assert len(files) == 2 and files == sorted(p.name for p in o2.iterdir())      # two threshold configurations
    for f in files:
        a, b = (o1 / f).read_text(), (o2 / f).read_text()
        assert a == b and a.count("\n") == 11


RAGGED = r"""
import sys
sys.path.insert(0, r"{repo}"); sys.path.insert(0, r"{repo}/tissue-model-analysis-tools_amd")
from tmat_amd import distributed
ws, rank, _ = distributed.init_process_group_from_env()
depth = [3, 0, 7, 1, 2]                                   # slices per stack: ragged rows per rank
rows = []
for si in distributed.shard_indices(len(depth), rank, ws):
    rows += [(int(si) * (1 << 20) + z, z % 2, 0.125 * z + int(si), 0.0) for z in range(depth[int(si)])]
allrows = distributed.gather_rows_ragged(rows)
expect = [(si * (1 << 20) + z, z % 2, 0.125 * z + si, 0.0) for si in range(len(depth)) for z in range(depth[si])]
assert allrows == expect, (rank, allrows, expect)
distributed.finish_process_group()
print("rank", rank, "ok", len(allrows))
"""


def test_gather_rows_ragged_world2(tmp_path):
    """the invasion-depth script's exchange: one row per Z slice, stacks of different depth on different ranks"""
    script = tmp_path / "ragged.py"
    script.write_text(RAGGED.format(repo=str(REPO)))
    assert _run(script, [], 2).count("ok") == 2
    assert _run(script, [], 1).count("ok") == 1


FAILING = r"""
import sys
sys.path.insert(0, r"{repo}"); sys.path.insert(0, r"{repo}/tissue-model-analysis-tools_amd")
import numpy as np
from tmat_amd import branches, distributed
ws, rank, _ = distributed.init_process_group_from_env()
ids = ["img%02d" % i for i in range(6)]
def load_fn(i):
    if i == "img04":                       # lies in the LAST rank's shard: the first rank finishes its work and waits in the gather
        print("cannot read", i, flush=True)
        raise branches.InputError(i)
    return np.full((8, 8), int(i[3:]), np.uint16)
def analyze_fn(b, w, thresh, input_bits):
    if "{mode}" == "analyze" and int(b.max()) >= 4:         # any exception out of the analyser (a HIP error, a shape surprise) on the last rank
        raise RuntimeError("analyser failed on the batch that holds image %d" % int(b.max()))
    return [(k, 1, 2.0, 3.0) for k in range(len(b))]
if "{mode}" == "analyze":
    load_fn = lambda i: np.full((8, 8), int(i[3:]), np.uint16)
try:
    branches.run_sharded(ids, load_fn, lambda i, im: 100.0, analyze_fn,
                         dict(graph_thresh_1=[5, 7]), rank, ws, log=lambda m: None)
except distributed.RankFailed:
    distributed.finish_process_group()
    print("rank", rank, "failed cleanly", flush=True)
    sys.exit(1)
print("rank", rank, "unexpectedly succeeded")
"""


@pytest.mark.parametrize("mode", ["load", "analyze"])
@pytest.mark.parametrize("nproc", [1, 2])
def test_a_rank_that_cannot_load_an_image_fails_the_whole_run_without_a_hang(tmp_path, nproc, mode):
    """ADVICE (round 2): a rank that sys.exit()s inside its shard loop leaves the others blocked in the all-gather; the failure
    now travels through the collective as a marker row and every rank exits with code 1.  Round 3's ADVICE: the same for ANY
    exception of the per-rank work ("analyze": the analyser raises on the last rank's batch), not only for an unreadable image."""
    script = tmp_path / "failing.py"
    script.write_text(FAILING.format(repo=str(REPO), mode=mode))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, str(script)] if nproc == 1 else [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
        "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=120)       # a hang would hit the timeout
    assert r.returncode != 0
    assert r.stdout.count("failed cleanly") == nproc and "unexpectedly" not in r.stdout, r.stdout + r.stderr
