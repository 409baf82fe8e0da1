"""bench.py's multi-rank tail (barrier, max-over-ranks of the elapsed time, the one row all-gather, the JSON line) on the CPU:
world_size 2 over gloo with the stand-in analyser of TMAT_BENCH_STUB=1, launched exactly as the driver launches the GPU run."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]


def _bench(nproc, images):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, TMAT_BENCH_STUB="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    args = [str(REPO / "bench.py"), "--gpus", str(nproc), "--steps", "2", "--warmup", "1", "--images", str(images)]
    if nproc == 1:
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout          # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("nproc", [1, 2])
def test_bench_json_line_under_gloo(nproc):
    images = 3
    out = _bench(nproc, images)
    assert out["n_gpus"] == nproc and out["steps"] == 2 and out["warmup"] == 1
    assert out["scaling"] == "weak" and out["higher_is_better"] is True and out["vs_baseline"] is None
    assert out["config"]["rows_gathered"] == nproc * images          # every rank's rows reached rank 0 through the one all-gather
    assert out["config"]["images_per_gpu"] == images
    assert out["cpu_baseline"] is None
    assert out["value"] > 0 and abs(out["value"] - nproc * images * 2 / (out["ms_per_step"] * 2e-3)) < 1e-3 * out["value"] + 1e-3
    assert "roofline" in out and out["roofline"]["traffic"] is None
    assert "TMAT_BENCH_STUB" in out["data"]


def test_bench_refuses_multi_gpu_without_launcher():
    env = dict(os.environ, TMAT_BENCH_STUB="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--images", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "torch.distributed.run" in (r.stdout + r.stderr)
