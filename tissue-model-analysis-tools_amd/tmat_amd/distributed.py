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


ROW_DTYPE = np.dtype([("index", "<i8"), ("count", "<i8"), ("total", "<f8"), ("avg", "<f8")])      # the 32-byte result row


def host_threads_per_rank(world_size: int) -> int:
    """host worker threads one rank may use for the thinning / DMT / MorseGraph stages: the cores this process may run
    on, divided by the ranks of the node (every rank otherwise spawns hardware_concurrency workers)"""
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 8
    return max(1, min(ncpu // max(1, world_size), 32))


class RankFailed(RuntimeError):
    """some rank of the run could not process its shard (the message is printed on that rank)"""


FAILED_INDEX = -2          # index of the marker row a failed rank contributes to the gather


def gather_rows(rows_local, n_total=None, device=None, failed=False):
    """rows_local: list of (index, count, total_px, avg_px).  Returns the rows of ALL ranks sorted by index (on every
    rank) with ONE fixed-size all-gather: every rank pads its shard to ceil(n_total / world) rows of 32 bytes (the
    padding carries index -1).  `n_total` = number of rows over all ranks (the images of the run); when it is not
    given, a contiguous block partition of at most len(rows_local) + 1 rows per rank is assumed (shard_indices).
    Single process: returns the input sorted.
    `failed`: this rank could not finish its shard (an unreadable image, say).  It must still enter the collective -- a rank
    that exits in front of it leaves the others blocked in the all-gather until the launcher kills them -- so it contributes
    one marker row (index FAILED_INDEX) and EVERY rank raises RankFailed after the gather."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        if failed:
            raise RankFailed("the run failed")
        return sorted(rows_local, key=lambda r: r[0])
    ws = dist.get_world_size()
    cap = max(1, -(-int(n_total) // ws) if n_total is not None else len(rows_local) + 1)
    if failed:
        rows_local = [(FAILED_INDEX, 0, 0.0, 0.0)]
    if len(rows_local) > cap:
        raise ValueError(f"gather_rows: {len(rows_local)} local rows exceed the shard capacity {cap}")
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    mine = np.zeros(cap, ROW_DTYPE)
    mine["index"] = -1
    for k, r in enumerate(rows_local):
        mine[k] = (int(r[0]), int(r[1]), float(r[2]), float(r[3]))
    inp = torch.from_numpy(mine.view(np.uint8).reshape(cap * ROW_DTYPE.itemsize)).to(dev)
    out = torch.empty(ws * cap * ROW_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, inp)           # the one collective of the path (RCCL over xGMI on GPUs)
    allrows = out.cpu().numpy().view(ROW_DTYPE)
    if (allrows["index"] == FAILED_INDEX).any():
        raise RankFailed("a rank of the run failed")
    allrows = allrows[allrows["index"] >= 0]
    return sorted(((int(r["index"]), int(r["count"]), float(r["total"]), float(r["avg"])) for r in allrows), key=lambda r: r[0])


def init_process_group_from_env():
    """(world_size, rank, device index); under torch.distributed.run the process group is created (nccl = RCCL when a GPU is
    visible, gloo otherwise) and this rank's device selected.
    TMAT_DIST_REHEARSE=1 (tests): more ranks than GPUs -- rank r uses device local_rank % device_count and the collectives run over
    gloo (RCCL refuses two ranks on one device); the scripts' whole multi-rank flow then runs on a one-GPU box with the real library."""
    ws, rank, local_rank = world()
    if ws > 1:
        import torch
        import torch.distributed as dist
        rehearse = os.environ.get("TMAT_DIST_REHEARSE") == "1" and torch.cuda.is_available()
        if rehearse:
            local_rank = local_rank % max(1, torch.cuda.device_count())
        if not dist.is_initialized():
            backend = "nccl" if (torch.cuda.is_available() and not rehearse) else "gloo"
            if torch.cuda.is_available():
                torch.cuda.set_device(local_rank)
            dist.init_process_group(backend)
    return ws, rank, local_rank


def finish_process_group():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def gather_rows_ragged(rows_local, device=None, failed=False):
    """gather_rows for shards whose row counts are not known up front (the invasion-depth tool: one row per Z slice, stacks of
    different depth): an all-reduce(max) of the local counts fixes the shard capacity, then the one all-gather"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        if failed:
            raise RankFailed("the run failed")
        return sorted(rows_local, key=lambda r: r[0])
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    cap = torch.tensor([max(1, len(rows_local))], dtype=torch.int64, device=dev)
    dist.all_reduce(cap, op=dist.ReduceOp.MAX)
    return gather_rows(rows_local, n_total=int(cap.item()) * dist.get_world_size(), device=device, failed=failed)
