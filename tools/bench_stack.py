#!/usr/bin/env python3
"""Measurement of the Z-stack (Sato) branch (reference scripts/compute_branches.py:224-306): synthetic
Z x 1024 x 1024 uint16 stacks through tmat_analyze_stack (host buffer in, one result row out: upload, per-slice gaussian,
resize to 384 wide, Sato over the slice pairs, the mask stages, DMT front end on the device, graph sweeps on the host).
Prints one JSON line in bench.py's format: value = stacks/s; cpu_baseline = oracle/sato.py (scipy.ndimage on the host) on
the same stacks.

    python tools/bench_stack.py [--slices 24] [--size 1024] [--steps 5] [--warmup 1] [--hessian gaussian_derivatives]
"""
import argparse
import json
import sys
import time
from pathlib import Path


REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tissue-model-analysis-tools_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slices", type=int, default=24)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--hessian", default="gaussian_derivatives", choices=["gaussian_derivatives", "gradient"])
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    from tmat_amd import _lib, branches, sato, synth
    h = _lib.Handle(None, 0)
    stacks = [synth.synth_stack(i, a.slices, a.size, a.size, n_vessels=30) for i in range(2)]
    cfg = {"graph_thresh_1": 5, "graph_thresh_2": 10, "graph_smoothing_window": 12, "min_branch_length": 12, "remove_isolated_branches": False}
    sw, mn, mx = branches.graph_px_params(cfg, 384, 1000.0)

    def step(i):
        return sato.analyze_stack(h, stacks[i % 2], 5, 10, sw, mn, mx, False, hessian=a.hessian)
    rows = [step(i) for i in range(max(a.warmup, 1))]
    t0 = time.perf_counter()
    for i in range(a.steps):
        rows.append(step(i))
    dt = time.perf_counter() - t0
    # stage split of one stack (each call uploads / downloads its own operands)
    t1 = time.perf_counter()
    vol = sato.stack_prepare(h, stacks[0], (384 * a.size // a.size, 384))
    t2 = time.perf_counter()
    field = sato.vessel_field(h, vol, a.hessian)
    t3 = time.perf_counter()
    sato.field_stats(h, field, 5, 10, sw, mn, mx, False)
    t4 = time.perf_counter()
    out = {"metric": "z_stacks_per_sec", "value": a.steps / dt, "unit": "stacks/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64 accumulate / f32 store",
           "data": "synthetic", "config": {"workload": f"Z-stack (Sato) branch: {a.slices} x {a.size} x {a.size} u16 -> 384-wide field -> row, host buffers in (PCIe inclusive)",
                                           "hessian": a.hessian},
           "stages_ms": {"stack_prepare": (t2 - t1) * 1e3, "vessel_field": (t3 - t2) * 1e3, "field_stats": (t4 - t3) * 1e3},
           "rows": [list(r) for r in rows[-2:]]}
    # roofline of the stage that holds the branch's arithmetic: scipy-exact separable correlations of the Sato Hessian, f64 accumulation
    # (two adds + one multiply per tap pair, no contraction): 10 passes per sigma, radius r(sigma) tap pairs per pixel and pass
    import math
    if a.hessian == "gaussian_derivatives":
        radii = [int((100.0 if sg <= 1 else 8.0) * sg / math.sqrt(2) + 0.5) for sg in sato.SATO_SIGMAS]
        pairs_px = 10 * sum(radii)
    else:
        pairs_px = 2 * sum(int(4.0 * sg + 0.5) for sg in sato.SATO_SIGMAS)        # gaussian smoothing only; the gradients are differences
    pairs = float(pairs_px) * field.shape[0] * field.shape[1] * (a.slices - 1)
    F64_PAIR_PEAK = 39.3e12 / 3.0        # MI355X f64 vector rate (lane-ops/s) over the 3 operations of a tap pair
    out["roofline"] = {"bound": "f64 vector ALU", "achieved": pairs / (t3 - t2) / 1e12, "peak": F64_PAIR_PEAK / 1e12, "unit": "10^12 tap pairs/s",
                       "frac": pairs / (t3 - t2) / F64_PAIR_PEAK, "traffic": None, "kernel": "tmat::corr1d_col_kernel / corr1d_row_kernel (Sato Hessian)",
                       "note": f"{pairs_px} tap pairs per pixel and slice pair over the WHOLE vessel_field call (upload of the prepared volume, unsharp, "
                               "canny, medial axis, region growing and mask filter inside the time); per kernel: profiles/*_stack_kernel_stats.csv"}
    if not a.no_cpu:
        from oracle import sato as osato
        c0 = time.perf_counter()
        orow = osato.analyze_stack(stacks[0], cfg, 1000.0, hessian=a.hessian)
        c1 = time.perf_counter()
        out["cpu_baseline"] = {"value": 1.0 / (c1 - c0), "unit": "stacks/s", "cores": 1, "kind": "port", "sample": "one stack through oracle/sato.py (scipy.ndimage, single thread)"}
        out["parity"] = list(orow) == list(rows[max(a.warmup, 1) - 1 if False else 0])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
