"""GPU: bench.py under TWO ranks on the one-GPU box (TMAT_BENCH_REHEARSE=1: both ranks use device 0, the two collectives run over
gloo because RCCL refuses two ranks on one device).  Everything an N-GPU run executes except the RCCL transport runs here with the
real library: torch and libtmat_hip.so in one process, per-rank handles and images, the barrier / max-over-ranks timing, the row
all-gather, the one JSON line.  The number it prints is not a measurement."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parents[1]


def test_two_ranks_on_one_gpu():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, TMAT_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    images = 16
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(REPO / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--images", str(images), "--max-patches", "400"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["rows_gathered"] == 2 * images and out["config"]["images_per_gpu"] == images
    assert out["config"]["copies_identical"] is True and out["value"] > 0
    assert out["cpu_baseline"] is None and out["alt"] is None
    assert "TMAT_BENCH_REHEARSE" in out["data"]
    assert out["roofline"]["frac"] > 0.3          # the dominant kernel really ran (HIP-event timing of its launches)
