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


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(img, weights, handle, log, n_sample_patches=40, repeats=3):
    """Reference CPU path = the oracle port (UNet on all usable cores + the oracle's numpy/C host stages), timed on a
    BOUNDED sample of one 1024x1024 image of the workload: all non-UNet stages in full, the UNet on `n_sample_patches`
    of the image's 200 patches (scaled up).  Every stage is timed `repeats` times and the MEDIAN counts.  Two CPU UNet
    implementations exist in the oracle (the as-written PyTorch-CPU graph, and oracle/unet_exact.c with OpenMP and the
    sub-pixel form); the faster one is the baseline, the other is reported (timed once: it is 3x slower)."""
    import statistics
    import torch
    from oracle import blend, dmt, morph, morse, pipeline, unet as ou
    from tmat_amd import _lib
    cores = ou.usable_cores()
    torch.set_num_threads(cores)
    log(f"timing the CPU baseline (oracle port, {cores} threads, median of {repeats}) on a bounded sample of one image")
    t = time.perf_counter

    def med(fn, n=repeats):
        ts = []
        for _ in range(n):
            t0 = t(); r = fn(); ts.append(t() - t0)
        return statistics.median(ts), r

    def pre():
        small = morph.lanczos4_resize_u16(img, morph.resized_shape(img.shape, 0.625))
        return morph.rescale_intensity(small, (0, 1)).astype(np.float32)
    t_pre, x = med(pre)
    # tiles of the first D4 orientation (the same patches the reference feeds to keras predict)
    pad = np.pad(x, 160, constant_values=x.min())
    tiles = np.array([pad[i:i + 320, j:j + 320] for i in range(0, 641, 160) for j in range(0, 641, 160)])

    def unet(fwd):
        done = 0
        while done < n_sample_patches:
            k = min(16, n_sample_patches - done)                  # INFERENCE_BATCH_SIZE = 16
            fwd(weights, tiles[(np.arange(k) + done) % len(tiles)])
            done += k
    ou.forward_exact(weights, tiles[:2])                          # warm the thread pool / page in the weights
    t_exact, _ = med(lambda: unet(ou.forward_exact))
    t_unet_exact = t_exact * (200.0 / n_sample_patches)
    ou.forward_torch(weights, tiles[:2])
    t_torch, _ = med(lambda: unet(ou.forward_torch), 1)
    t_unet_torch = t_torch * (200.0 / n_sample_patches)
    log(f"  cpu baseline: UNet per image {t_unet_torch:.2f}s (PyTorch-CPU, as written, 1 run) / {t_unet_exact:.2f}s (oracle C, OpenMP, median)")
    t_unet = min(t_unet_torch, t_unet_exact)                  # the baseline is the faster CPU implementation
    t_blend, _ = med(lambda: blend.predict_img_with_smooth_windowing(x, 320, 2, lambda b, verbose=0: np.asarray(b)[..., None]))
    # downstream stages on the probability map of the same image (taken from the GPU path)
    pred = np.empty((1,) + x.shape, np.float64)
    _lib.check(_lib.lib().tmat_segment_batch(handle.raw, _lib.ptr(np.ascontiguousarray(img[None])), 1, img.shape[0], img.shape[1],
                                             0.625, _lib.ptr(pred)), "segment")

    def post():
        field, _, _ = morph.postprocess(pred[0], morph.dsamp_shape(img.shape, 384))
        f255 = morph.rescale_intensity(field, (0, 255))
        V, E = dmt.compute_dmt_graph(f255, 5.0, 10.0)
        sw, mn, mx = pipeline.px_params(CFG, 384, IMAGE_WIDTH_MICRONS)
        return morse.morse_stats(V, E, field.shape, sw, mn, mx, False, None)[1]
    t_post, n0 = med(post)
    total = t_pre + t_unet + t_blend + t_post
    log(f"  cpu baseline: pre {t_pre:.2f}s unet(scaled) {t_unet:.2f}s blend {t_blend:.2f}s post+graph {t_post:.2f}s")
    return {"value": round(1.0 / total, 5), "unit": "images/s", "cores": cores, "kind": "port",
            "kind_detail": "port (oracle/), 1-image sample",
            "sample": f"1-image sample: one 1024x1024 image, every stage the median of {repeats} runs: all non-UNet stages in full; UNet on "
                      f"{n_sample_patches} of its 200 patches (time scaled by 200/{n_sample_patches}) with the faster of the oracle's two CPU implementations",
            "cpu_model": _cpu_model(), "os_cpu_count": os.cpu_count(), "torch_threads": int(torch.get_num_threads()),
            "seconds_per_image": round(total, 2), "unet_seconds_per_image": round(t_unet, 2),
            "unet_seconds_per_image_torch_as_written": round(t_unet_torch, 2),
            "unet_seconds_per_image_oracle_c": round(t_unet_exact, 2), "count": int(n0)}


class _StubHandle:
    """TMAT_BENCH_STUB=1: the handle methods bench.py calls, without a GPU"""
    def prof_enable(self, on):
        pass

    def prof_read(self, reset=False):
        return 0.0, 0, 0.0

    def close(self):
        pass


def _gen_image(i):
    from tmat_amd import synth
    return synth.synth_image(i, SIZE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--images", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--distinct", type=int, default=0, help="distinct synthetic images generated (0 = all of them, SURVEY 8d; fewer are tiled)")
    ap.add_argument("--max-patches", type=int, default=1600, help="UNet patches resident per pass (200 per image)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the opt-in split-precision (bf16x3 / bf16x6) lines")
    args = ap.parse_args()

    t_start = time.perf_counter()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 8
    os.environ.setdefault("TMAT_HOST_THREADS", str(max(1, min(ncpu // max(1, world), 32))))      # = distributed.host_threads_per_rank
    # the CPU baseline gets the same core share as the GPU path's host stages (oracle/unet.py:usable_cores reads this)
    os.environ.setdefault("TMAT_ORACLE_THREADS", os.environ["TMAT_HOST_THREADS"])

    # synthetic inputs (image i of rank r from RandomState(1234 + r * 100003 + i)), generated by a process pool
    # BEFORE anything touches the GPU (fork after HIP initialisation is not safe)
    n_img = args.images
    n_distinct = n_img if args.distinct <= 0 else min(args.distinct, n_img)
    t_gen = time.perf_counter()
    workers = max(1, min(ncpu // max(1, world), 32, n_distinct))
    if workers > 1:
        import multiprocessing as mp
        # close() + join(), not the context manager: its exit calls terminate(), which SIGTERMs the workers -- under rocprofv3
        # every forked worker carries the profiler's signal handler and logs that as an abort
        pool = mp.get_context("fork").Pool(workers)
        try:
            distinct = pool.map(_gen_image, [rank * 100003 + i for i in range(n_distinct)], chunksize=1)
        finally:
            pool.close()
            pool.join()
    else:
        distinct = [_gen_image(rank * 100003 + i) for i in range(n_distinct)]
    t_gen = time.perf_counter() - t_gen

    import torch
    # TMAT_BENCH_STUB=1 (tests/test_bench_multirank.py): no GPU and no library -- a stand-in analyser computes the rows from the
    # pixels on the CPU, the process group is gloo, and everything AROUND the hot path (sharding, barriers, the max-over-ranks
    # of the elapsed time, the row all-gather, the JSON line) runs exactly as it does on GPUs.  Its number is not a measurement.
    stub = os.environ.get("TMAT_BENCH_STUB") == "1"
    # TMAT_BENCH_REHEARSE=1 (tests/test_gpu_bench_rehearsal.py): the real handles and kernels under N > 1 ranks on a box with FEWER GPUs than
    # ranks -- rank r uses device r % device_count and the collectives run over gloo (RCCL refuses two ranks on one device).  Everything
    # but the transport of the two collectives is what an N-GPU run executes; its number is not a measurement and the line says so.
    rehearse = os.environ.get("TMAT_BENCH_REHEARSE") == "1" and not stub
    dev_index = local_rank
    dist = None
    if world > 1:
        import torch.distributed as dist
        if stub:
            dist.init_process_group("gloo")
        elif rehearse:
            dev_index = local_rank % max(1, torch.cuda.device_count())
            torch.cuda.set_device(dev_index)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from tmat_amd import branches, synth

    host = np.stack([distinct[i % len(distinct)] for i in range(n_img)])
    if stub:
        weights = None
        handle = _StubHandle()
        dptr = L = None

        def step(n=n_img):
            return [(rank * n_img + i, int(host[i].sum() % 97), float(host[i].mean()), float(host[i].std())) for i in range(n)]
        step_host = step

        def handle_sync():
            pass
    else:
        from tmat_amd import _lib
        weights = synth.synth_weights(0)
        handle = _lib.Handle(synth.pack_weights(weights), dev_index, args.max_patches)
        L = _lib.lib()
        # resident in HBM before the timed region (each rank has its own images)
        dptr = ctypes.c_void_p()
        _lib.check(L.tmat_dev_alloc(handle.raw, host.nbytes, ctypes.byref(dptr)), "dev_alloc")
        _lib.check(L.tmat_dev_upload(handle.raw, dptr, _lib.ptr(host), host.nbytes), "dev_upload")

        def step(n=n_img):
            return branches.analyze_batch(handle, (n, SIZE, SIZE), CFG, IMAGE_WIDTH_MICRONS, first_index=rank * n_img,
                                          dev_ptr=dptr.value)

        def step_host(n=n_img):
            # SURVEY 8d's metric as written: the uint16 arrays are in HOST memory when the clock starts (tmat_analyze_batch: H2D inside)
            return branches.analyze_batch(handle, host[:n], CFG, IMAGE_WIDTH_MICRONS, first_index=rank * n_img)

        def handle_sync():
            _lib.check(L.tmat_sync(handle.raw), "sync")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize() if torch.cuda.is_available() else None
        handle_sync()

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:7.1f}s] {msg}", file=sys.stderr, flush=True)

    log(f"{n_distinct} distinct images generated in {t_gen:.1f}s by {workers} processes; inputs resident in HBM ({n_img} images/GPU), "
        f"starting {args.warmup} warmup step(s)")
    for _ in range(args.warmup):
        rows = step()
    log("warmup done, timing")
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rows = step()
    barrier()
    elapsed = time.perf_counter() - t0
    log(f"timed region {elapsed:.2f} s")
    # second figure, outside the contract's timed region: the same step through the host-pointer entry (PCIe inclusive), a few steps
    n_host_steps = max(1, min(2, args.steps))
    barrier()
    t0 = time.perf_counter()
    for _ in range(n_host_steps):
        host_rows = step_host()
    barrier()
    elapsed_host = time.perf_counter() - t0
    if [r[1:] for r in host_rows] != [r[1:] for r in rows]:
        raise SystemExit("bench: the host-pointer entry and the device-pointer entry produced different rows")
    log(f"host-input steps: {n_host_steps} in {elapsed_host:.2f} s")
    # HIP-event timing of the dominant kernel in a SEPARATE pass after the timed region (same kernels, same 1600-patch
    # launches: two passes of 8 images), on the stream the kernels are launched on.  Its rows double as a determinism /
    # race screen: the same images in another position of another run must give identical rows; tiled copies likewise.
    n_prof = min(16, n_img)
    handle.prof_enable(True)
    handle.prof_read(True)
    prof_rows = step(n_prof)
    conv_ms, conv_launches, conv_flops = handle.prof_read(True)
    handle.prof_enable(False)
    for i in range(n_prof):
        if prof_rows[i][1:] != rows[i][1:]:
            raise SystemExit(f"bench: image {i} gave different rows in two runs: {prof_rows[i]} vs {rows[i]}")
    nd = len(distinct)
    for i, r in enumerate(rows):
        if r[1:] != rows[i % nd][1:]:
            raise SystemExit(f"bench: image {i} and its copy {i % nd} produced different rows: {r} vs {rows[i % nd]}")

    if dist is not None:
        t = torch.tensor([elapsed, elapsed_host], dtype=torch.float64, device="cpu" if (stub or rehearse) else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, elapsed_host = float(t[0].item()), float(t[1].item())
        # the one collective of the path: a single fixed-size all-gather of the 32-byte result rows over RCCL/xGMI
        from tmat_amd import distributed
        n_rows = len(distributed.gather_rows(rows, n_total=n_img * world))
    else:
        n_rows = len(rows)

    out = None
    if rank == 0:
        total_images = n_img * world * args.steps
        value = total_images / elapsed
        achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        # HBM traffic of the dominant kernel comes from the committed PMC summary of the same pipeline (separate
        # rocprofv3 --pmc passes, tools/profile_round.sh); it is per launch of 1600 patches, like `achieved`
        traffic = None
        top_by_time = None
        traffic_source = None
        summ = sorted((REPO / "profiles").glob("r*_summary.json"))
        if summ and args.max_patches == 1600 and not stub:
            sj = json.loads(summ[-1].read_text())
            traffic = sj.get("traffic_bytes_per_launch")
            top_by_time = sj.get("top_kernels_by_total_time")
            traffic_source = f"profiles/{summ[-1].name} (committed rocprofv3 --pmc / --kernel-trace passes of this command; not re-measured in this run)"
        # path-level figure: every FLOP the step runs on the matrix cores (all 3x3 / sub-pixel / 1x1 layers) over the
        # whole step time (which also holds the vector-ALU layers, morphology, blending and the host stages)
        path_flops = synth.mfma_flops_per_patch() * 200.0 * n_img * world * args.steps
        path_tflops = path_flops / elapsed / 1e12 / world
        out = {
            "metric": "images/sec (1024x1024 uint16) through compute_branches",
            "value": round(value, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "value_host_inputs": round(n_img * world * n_host_steps / elapsed_host, 4),
            "value_host_inputs_note": f"SURVEY 8d's metric as written -- uint16 arrays in HOST memory when the clock starts (tmat_analyze_batch, "
                                      f"2 MB/image of H2D inside the timed region), {n_host_steps} step(s) after the contract's timed region; "
                                      "`value` is the contract's figure (inputs resident in HBM)",
            "config": {"workload": f"{n_img} synthetic 1024x1024 uint16 Z-projections per GPU ({len(distinct)} distinct), inputs pre-resident "
                                   "in HBM (host-pointer entry adds 2 MB/image of H2D), tiled UNet seg (200 patches/image, random-init "
                                   "structured weights) + DMT branch extraction, default_branching_computation.json",
                       "images_per_gpu": n_img, "distinct_images": len(distinct), "patches_per_image": 200,
                       "rows_gathered": n_rows, "copies_identical": True, "host_threads": int(os.environ["TMAT_HOST_THREADS"])},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "tmat::conv_mfma_kernel<128, 128, 4, 2, 3, false> (3x3 implicit-GEMM on v_mfma_f32_32x32x2_f32; "
                                   "4 of the 8 transposed-conv layers -- both convolutions of up block 0 and the second one of blocks 1 and 2 --, "
                                   "52 % of the 3x3 / sub-pixel MFMA FLOPs; the largest kernel by total time)",
                         "launches": int(conv_launches), "avg_launch_ms": round(conv_ms / max(conv_launches, 1), 4),
                         "timed_in": f"separate pass of {n_prof} images after the timed region (HIP events on the launch stream)",
                         "path_achieved": round(path_tflops, 2), "path_frac": round(path_tflops / MFMA_F32_PEAK_TFLOPS, 4),
                         "path_note": "executed MFMA FLOPs of the whole step (19.08 GFLOP/patch: every 3x3 / sub-pixel / 1x1 layer) / ms_per_step, per GPU",
                         "top_kernels_by_total_time": top_by_time, "top_kernels_source": traffic_source},
        }
        sample = [r for r in rows[:4]]
        out["config"]["sample_rows"] = [[int(r[0]), int(r[1]), round(r[2], 3)] for r in sample]

    # opt-in split-precision lines (tmat_set_precision("bf16x3" | "bf16x6")): never the headline; one full step of the same images
    # per mode, its rows checked against the f32 rows with the north-star tolerance (counts equal, lengths within 1e-4 relative)
    # and REPORTED either way
    if rank == 0 and world == 1 and not stub and not args.no_alt:
        BF16_PEAK_TFLOPS = 2500.0                        # MI355X_MICROARCH.md: dense bf16 MFMA
        out["alt"] = []
        for mode, nprod, what in (("bf16x3", 3, "bf16 hi/lo split of the f32 operands, 3 bf16 MFMA products per MAC"),
                                  ("bf16x6", 6, "bf16 hi/mid/lo split (the whole 24-bit mantissa), 6 bf16 MFMA products per MAC")):
            handle.set_precision(mode)
            step(n_prof)                                 # warm-up (the split weight copies are made on the first switch)
            handle_sync()
            t0 = time.perf_counter()
            alt_rows = step()
            handle_sync()
            alt_elapsed = time.perf_counter() - t0
            handle.prof_enable(True)
            handle.prof_read(True)
            step(n_prof)
            alt_ms, alt_launches, alt_flops = handle.prof_read(True)
            handle.prof_enable(False)
            bad = [(int(a[0]), int(a[1]), int(b[1])) for a, b in zip(rows, alt_rows) if a[1] != b[1]]
            rel = [abs(b[2] - a[2]) / max(abs(a[2]), 1e-30) for a, b in zip(rows, alt_rows) if a[1] == b[1] and a[1] > 0]
            max_rel = max(rel) if rel else 0.0
            n_same = sum(1 for a, b in zip(rows, alt_rows) if a[1:] == b[1:])
            alt_ach = alt_flops / (alt_ms * 1e-3) / 1e12 if alt_ms > 0 else 0.0
            out["alt"].append({
                "precision": mode, "dtype": what + ", f32 accumulate (sepconv / stem / final stay f32)",
                "value": round(n_img / alt_elapsed, 4), "unit": "images/s", "ms_per_step": round(alt_elapsed * 1e3, 3),
                "speedup_vs_f32": round((n_img / alt_elapsed) / value, 3),
                "parity_vs_f32": {"images": n_img, "rows_bit_identical": n_same, "count_mismatches": len(bad), "mismatching_rows": bad[:8],
                                  "max_rel_length_diff_where_counts_agree": float(f"{max_rel:.3e}"),
                                  "tolerance": "counts equal, lengths within 1e-4 relative (north_star)",
                                  "pass": bool(not bad and max_rel <= 1e-4)},
                "roofline": {"bound": "mfma", "kernel": f"tmat::conv_mfma_kernel<128, 128, 4, 2, 3, false, {nprod // 3}> (the dominant kernel's split-precision instantiation)",
                             "achieved": round(alt_ach, 2), "executed": round(nprod * alt_ach, 2), "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(nprod * alt_ach / BF16_PEAK_TFLOPS, 4), "launches": int(alt_launches),
                             "avg_launch_ms": round(alt_ms / max(alt_launches, 1), 4),
                             "note": f"achieved = algorithmic FLOPs (2 M N K) / HIP-event time; executed = {nprod} x achieved ({nprod} bf16 products per MAC) is what the bf16 peak bounds"},
            })
            log(f"alt {mode}: {out['alt'][-1]['value']} images/s, parity pass {out['alt'][-1]['parity_vs_f32']['pass']} "
                f"({len(bad)} count mismatches, {n_same} of {n_img} rows bit-identical)")
        handle.set_precision("f32")
    elif rank == 0:
        out["alt"] = None

    if rank == 0 and world == 1 and not args.no_cpu_baseline and not stub:
        out["cpu_baseline"] = cpu_baseline(host[0], weights, handle, log)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0 and stub:
        out["data"] = "synthetic (TMAT_BENCH_STUB: stand-in analyser on the CPU, not a measurement)"
    if rank == 0 and rehearse:
        out["data"] = f"synthetic (TMAT_BENCH_REHEARSE: {world} ranks share {torch.cuda.device_count()} GPU(s), collectives over gloo -- not a measurement)"

    if not stub:
        _lib.check(L.tmat_dev_free(handle.raw, dptr), "dev_free")
    handle.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
