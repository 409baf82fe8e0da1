#!/usr/bin/env python3
"""bench.py -- images/sec (1024x1024 uint16) through the compute_branches hot path on N MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the whole hot path (Lanczos4 + rescale, 200-patch tiled UNet with D4 TTA and
spline blending, mask filtering / medial axis / EDT / resize, DMT graph, MorseGraph statistics) over a
batch of `--images` (default 256 = BASELINE.json configs[1]) synthetic 1024x1024 uint16 images PER
GPU, with the inputs already resident in HBM.  Images are independent, so ranks shard them with no
data-path collective (weak scaling); the result rows are all-gathered over RCCL once at the end.
Prints ONE JSON line on rank 0 (see README / DESIGN.md "Measurement").
"""
import argparse
import ctypes
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
for p in (REPO, REPO / "tissue-model-analysis-tools_amd", REPO / "tools"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import numpy as np  # noqa: E402

CFG = dict(graph_thresh_1=5, graph_thresh_2=10, graph_smoothing_window=12, min_branch_length=12,
           remove_isolated_branches=False)           # config/default_branching_computation.json
IMAGE_WIDTH_MICRONS = 1000.0
SIZE = 1024
MFMA_F32_PEAK_TFLOPS = 157.3                          # MI355X_MICROARCH.md: f32-input MFMA = vector f32 peak


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--images", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic images generated (tiled to --images)")
    ap.add_argument("--max-patches", type=int, default=1600, help="UNet patches resident per pass (200 per image)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    t_start = time.perf_counter()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    os.environ.setdefault("TMAT_HOST_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, world))))

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from tmat_amd import _lib, branches, synth

    weights = synth.synth_weights(0)
    handle = _lib.Handle(synth.pack_weights(weights), local_rank, args.max_patches)
    L = _lib.lib()

    # synthetic inputs, resident in HBM before the timed region (each rank gets its own images)
    n_img = args.images
    distinct = [synth.synth_image(rank * 100003 + i, SIZE) for i in range(min(args.distinct, n_img))]
    host = np.stack([distinct[i % len(distinct)] for i in range(n_img)])
    dptr = ctypes.c_void_p()
    _lib.check(L.tmat_dev_alloc(handle.raw, host.nbytes, ctypes.byref(dptr)), "dev_alloc")
    _lib.check(L.tmat_dev_upload(handle.raw, dptr, _lib.ptr(host), host.nbytes), "dev_upload")

    def step():
        return branches.analyze_batch(handle, (n_img, SIZE, SIZE), CFG, IMAGE_WIDTH_MICRONS, first_index=rank * n_img,
                                      dev_ptr=dptr.value)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize() if torch.cuda.is_available() else None
        handle_sync()

    def handle_sync():
        _lib.check(L.tmat_sync(handle.raw), "sync")

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:7.1f}s] {msg}", file=sys.stderr, flush=True)

    log(f"inputs resident in HBM ({n_img} images/GPU), starting {args.warmup} warmup step(s)")
    for _ in range(args.warmup):
        rows = step()
    log("warmup done, timing")
    handle.prof_enable(True)
    handle.prof_read(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rows = step()
    barrier()
    elapsed = time.perf_counter() - t0
    log(f"timed region {elapsed:.2f} s")
    conv_ms, conv_launches, conv_flops = handle.prof_read(True)
    handle.prof_enable(False)

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the one collective of the path: all-gather of the 32-byte result rows over RCCL/xGMI
        mine = torch.tensor([[r[0], r[1], r[2], r[3]] for r in rows], dtype=torch.float64, device="cuda")
        allrows = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allrows, mine)
        n_rows = sum(int(a.shape[0]) for a in allrows)
    else:
        n_rows = len(rows)

    out = None
    if rank == 0:
        total_images = n_img * world * args.steps
        value = total_images / elapsed
        achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        out = {
            "metric": "images/sec (1024x1024 uint16) through compute_branches",
            "value": round(value, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n_img} synthetic 1024x1024 uint16 Z-projections per GPU ({len(distinct)} distinct, tiled), "
                                   "tiled UNet seg (200 patches/image, random-init structured weights) + DMT branch extraction, "
                                   "default_branching_computation.json", "images_per_gpu": n_img, "patches_per_image": 200,
                       "rows_gathered": n_rows, "host_threads": int(os.environ["TMAT_HOST_THREADS"])},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                         "kernel": "conv_mfma_kernel (3x3 implicit-GEMM, v_mfma_f32_32x32x2_f32)",
                         "launches": int(conv_launches), "avg_launch_ms": round(conv_ms / max(conv_launches, 1), 4)},
        }
        sample = [r for r in rows[:4]]
        out["config"]["sample_rows"] = [[int(r[0]), int(r[1]), round(r[2], 3)] for r in sample]

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # reference CPU path = the oracle port with all-core PyTorch-CPU convolutions, on a bounded sample
        from oracle import pipeline
        torch.set_num_threads(os.cpu_count() or 8)
        log("timing the CPU baseline (oracle port, PyTorch-CPU UNet) on one image")
        t0 = time.perf_counter()
        n0, tot0, avg0 = pipeline.analyze_image(host[0], weights, CFG, IMAGE_WIDTH_MICRONS, unet_kind="torch")
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(1.0 / dt, 5), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "1 of the 1024x1024 images (6.2 TFLOP of convolutions), oracle pipeline with PyTorch-CPU UNet",
                               "count_cpu": int(n0), "count_gpu": int(rows[0][1])}
    elif rank == 0:
        out["cpu_baseline"] = None

    _lib.check(L.tmat_dev_free(handle.raw, dptr), "dev_free")
    handle.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
