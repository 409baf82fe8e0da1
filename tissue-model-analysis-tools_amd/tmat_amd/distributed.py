"""Image-per-GPU sharding of a run.  Images are independent
(reference compute_branches.py:585-594 loops over them), so ranks take disjoint index sets with no
data-path collective; the only exchange is one all-gather of the 32-byte result rows at the end
(RCCL over xGMI when the backend is "nccl", gloo in the CPU tests)."""
from __future__ import annotations

import os

import numpy as np


def world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_indices(n: int, rank: int, world_size: int) -> np.ndarray:
    """contiguous block partition: rank r gets [r*n/world, (r+1)*n/world)"""
    lo = (n * rank) // world_size
    hi = (n * (rank + 1)) // world_size
    return np.arange(lo, hi)


def gather_rows(rows_local, device=None):
    """rows_local: list of (index, count, total_px, avg_px).  Returns the rows of ALL ranks sorted by index
    (on every rank).  Single process: returns the input sorted."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return sorted(rows_local, key=lambda r: r[0])
    ws = dist.get_world_size()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    cnt = torch.tensor([len(rows_local)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(cnt) for _ in range(ws)]
    dist.all_gather(counts, cnt)
    m = max(int(c.item()) for c in counts)
    buf = torch.zeros((m, 4), dtype=torch.float64, device=dev)
    if rows_local:
        buf[: len(rows_local)] = torch.tensor([[float(r[0]), float(r[1]), r[2], r[3]] for r in rows_local], dtype=torch.float64)
    allbuf = [torch.zeros_like(buf) for _ in range(ws)]
    dist.all_gather(allbuf, buf)
    out = []
    for c, b in zip(counts, allbuf):
        for r in b[: int(c.item())].cpu().tolist():
            out.append((int(r[0]), int(r[1]), r[2], r[3]))
    return sorted(out, key=lambda r: r[0])
