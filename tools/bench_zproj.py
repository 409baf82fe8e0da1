#!/usr/bin/env python3
"""Measurement of the next-row workload (SURVEY.md §8f-2 / BASELINE config #3): focus-stacking Z projection of
synthetic 16 x 2048 x 2048 uint16 stacks, resident in HBM.  Prints one JSON line in bench.py's format:
value = stacks/s, roofline = algorithmic HBM bytes (Z H W 2 in + H W 2 out per stack) / kernel time against 8 TB/s,
cpu_baseline = the numpy oracle on a bounded crop of one stack, scaled by area.

    python tools/bench_zproj.py [--stacks 32] [--steps 5] [--warmup 2]
"""
import argparse
import ctypes as C
import json
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tissue-model-analysis-tools_amd"))


def synth_stack(i, Z=16, S=2048):
    """vessel image of bench.py's generator per slice, defocused by |z - z0|, built small and tiled"""
    from scipy import ndimage as ndi
    from tmat_amd import synth
    base = synth.synth_image(i, 512, 12, scale=1.0).astype(np.float64)
    z0 = (i * 5) % Z
    sl = [np.clip(ndi.gaussian_filter(base, abs(z - z0) * 0.7) if z != z0 else base, 0, 65535).astype(np.uint16) for z in range(Z)]
    return np.tile(np.stack(sl), (1, S // 512, S // 512))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stacks", type=int, default=32)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chain", action="store_true",
                    help="BASELINE config #3 end to end: project on the device, then analyse the 2048x2048 projections "
                         "(648 patches each) with the branching pipeline without leaving HBM; prints a second JSON line")
    a = ap.parse_args()
    from tmat_amd import _lib
    from oracle import zproj as oz
    L = _lib.lib()
    h = _lib.Handle(None, 0)
    Z, S = 16, 2048
    distinct = [synth_stack(i, Z, S) for i in range(4)]
    per_in, per_out = Z * S * S * 2, S * S * 2
    din, dout = C.c_void_p(), C.c_void_p()
    _lib.check(L.tmat_dev_alloc(h.raw, a.stacks * per_in, C.byref(din)), "alloc")
    _lib.check(L.tmat_dev_alloc(h.raw, a.stacks * per_out, C.byref(dout)), "alloc")
    for i in range(a.stacks):
        st = distinct[i % 4]
        _lib.check(L.tmat_dev_upload(h.raw, C.c_void_p(din.value + i * per_in), st.ctypes.data_as(C.c_void_p), per_in), "upload")

    def step():
        _lib.check(L.tmat_zproj_dev(h.raw, din, a.stacks, Z, S, S, 0, dout), "zproj")
        _lib.check(L.tmat_sync(h.raw), "sync")
    for _ in range(a.warmup):
        step()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    dt = (time.perf_counter() - t0) / a.steps
    # parity spot check of the timed output against the oracle on a crop
    out = np.empty((S, S), np.uint16)
    import ctypes
    hip_out = (C.c_uint16 * (S * S)).from_buffer(out)
    # read back stack 0 through the host entry point (same kernel) to compare
    got = h.zproj(distinct[0][None], "fs")[0]
    crop = distinct[0][:, :256, :256]
    assert np.array_equal(got[:252, :252], oz.proj_focus_stacking(crop)[:252, :252]), "GPU result differs from the oracle"
    # CPU baseline: oracle on a 512 x 512 crop of one stack (all 16 slices), scaled by area
    t1 = time.perf_counter()
    oz.proj_focus_stacking(distinct[1][:, :512, :512])
    cpu_s = (time.perf_counter() - t1) * (S / 512) ** 2
    alg = a.stacks * (per_in + per_out)
    print(json.dumps({
        "metric": "stacks/sec (16x2048x2048 uint16) through compute_zproj -m fs", "value": round(a.stacks / dt, 3), "unit": "stacks/s",
        "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
        "config": {"workload": f"{a.stacks} synthetic 16x2048x2048 uint16 Z stacks resident in HBM, focus stacking (BASELINE config #3, projection stage)"},
        "roofline": {"bound": "hbm", "achieved": round(alg / dt / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                     "frac": round(alg / dt / 8e12, 4), "traffic": None, "kernel": "tmat::zproj_focus_kernel"},
        "cpu_baseline": {"value": round(1.0 / cpu_s, 4), "unit": "stacks/s", "cores": 1, "kind": "port",
                         "sample": "oracle/zproj.py (numpy) on a 512x512 crop of one stack, time scaled by 16"}}))


def chain(stacks=8):
    """zproj -> compute_branches on device-resident stacks (config #3): images/s of the whole chain"""
    from tmat_amd import _lib, branches, synth
    L = _lib.lib()
    h = _lib.Handle(synth.pack_weights(synth.synth_weights(0)), 0, 1944)       # 3 images of 648 patches per pass
    Z, S = 16, 2048
    per_in = Z * S * S * 2
    din, dproj = C.c_void_p(), C.c_void_p()
    _lib.check(L.tmat_dev_alloc(h.raw, stacks * per_in, C.byref(din)), "alloc")
    _lib.check(L.tmat_dev_alloc(h.raw, stacks * S * S * 2, C.byref(dproj)), "alloc")
    for i in range(stacks):
        st = synth_stack(i % 2, Z, S)
        _lib.check(L.tmat_dev_upload(h.raw, C.c_void_p(din.value + i * per_in), st.ctypes.data_as(C.c_void_p), per_in), "upload")
    cfg = dict(graph_thresh_1=5, graph_thresh_2=10, graph_smoothing_window=12, min_branch_length=12)

    def step():
        _lib.check(L.tmat_zproj_dev(h.raw, din, stacks, Z, S, S, 0, dproj), "zproj")
        return branches.analyze_batch(h, (stacks, S, S), cfg, 2000.0, dev_ptr=dproj.value)
    step()
    t0 = time.perf_counter()
    rows = step()
    dt = time.perf_counter() - t0
    print(json.dumps({
        "metric": "stacks/sec (16x2048x2048 uint16) through compute_zproj -m fs + compute_branches", "value": round(stacks / dt, 3),
        "unit": "stacks/s", "n_gpus": 1, "steps": 1, "warmup": 1, "ms_per_step": round(dt * 1e3, 1), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{stacks} synthetic 16x2048x2048 uint16 Z stacks in HBM: focus stacking, then tiled UNet seg "
                               "(648 patches/image) + DMT branch extraction (BASELINE config #3)",
                   "sample_rows": [[r[0], r[1], round(r[2], 3)] for r in rows[:3]]}}))


if __name__ == "__main__":
    if "--chain" in sys.argv:
        chain()
    main()
