"""CPU, world_size 2 over gloo: the N>1 path (shard -> per-rank rows -> one all-gather) returns the
same row set as a single process, on every rank, for even and ragged shard sizes."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]

WORKER = r'''
import os, sys, json
sys.path.insert(0, r"{repo}"); sys.path.insert(0, r"{repo}/tissue-model-analysis-tools_amd")
import torch.distributed as dist
from tmat_amd import distributed
dist.init_process_group("gloo")
ws, rank, _ = distributed.world()
n = int(sys.argv[1])
mine = distributed.shard_indices(n, rank, ws)
rows = [(int(i), int(i) * 3 % 17, float(i) * 1.25, float(i) / 7.0) for i in mine]      # stand-in for analyze_batch rows
allrows = distributed.gather_rows(rows)
expect = [(i, i * 3 % 17, i * 1.25, i / 7.0) for i in range(n)]
assert allrows == expect, (rank, allrows[:3], expect[:3])
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok", len(allrows))
'''


@pytest.mark.parametrize("n", [0, 5, 8])
def test_gather_rows_world2(tmp_path, n):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(repo=str(REPO)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script), str(n)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2
