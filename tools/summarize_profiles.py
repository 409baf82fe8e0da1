#!/usr/bin/env python3
"""Turn gpurun_out/profiles_<round>/ (tools/profile_round.sh) into the committed profiles/<round>_* files:
the rocprofv3 --kernel-trace --stats table, the bench lines, and a JSON summary with the dominant kernel's
average duration and its per-launch HBM traffic from the PMC passes (FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE as read; both are in KiB)."""
import csv
import json
import shutil
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
DOM = "conv_mfma_kernel<128, 128, 4, 2, 3, false>"


def is_dom(name):
    """the f32 instantiation of the dominant kernel (the template has a trailing precision parameter since round 3: ', 0>')"""
    return "conv_mfma_kernel<128, 128, 4, 2, 3, false>" in name or "conv_mfma_kernel<128, 128, 4, 2, 3, false, 0>" in name


def main(rnd):
    src = REPO / "gpurun_out" / f"profiles_{rnd}"
    dst = REPO / "profiles"
    dst.mkdir(exist_ok=True)
    for name in ("kernel_stats.csv", "bench.json", "bench.err", "bench_under_rocprof.json"):
        if (src / name).exists():
            shutil.copy(src / name, dst / f"{rnd}_{name}")
    out = {"round": rnd, "dominant_kernel": "tmat::" + DOM}
    rows = list(csv.DictReader(open(src / "kernel_stats.csv")))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    # kernels of the low-priority side stream (ordered thinning, finish stage, DMT front end: pipeline.cpp:run_pass_host): they run
    # under the next pass's network, so their SUMMED duration is mostly time spent waiting for CU slots, not work
    side = ("ma_round_kernel", "ma_keys_kernel", "ma_dep_kernel", "ma_", "ms_hist_kernel", "ms_scan_kernel", "ms_scatter_kernel", "dmt_keys_kernel", "dmt_levels_kernel", "dmt_prep_kernel",
            "dmt_sort_kernel", "invert_u8", "weight_kernel", "gauss_axis", "zoom_clip", "rescale255", "minmax_kernel")

    # kernels of the tail of a pass (blend, threshold, mask filter, EDT: pipeline.cpp:enqueue_back), on the second stream beside the next
    # pass's network since round 3: their summed duration includes queueing for CU slots too
    tail = ("blend_kernel", "binarize_kernel", "threshold_kernel", "median13_kernel", "ccl_", "region_stats_kernel", "zhang_", "skel_fork_kernel",
            "decide_kernel", "apply_drop_kernel", "edt_", "thin_count")

    def stream_of(name):
        base = name.replace("void ", "").split("(")[0].split("<")[0].replace("tmat::", "")
        if any(base.startswith(k) for k in side):
            return "side (low priority; summed duration = mostly queueing under the next pass's network)"
        if any(base.startswith(k) for k in tail):
            return "second (tail of a pass beside the next pass's network; summed duration includes queueing)"
        return "main"
    out["top_kernels_by_total_time"] = [
        {"kernel": r["Name"].split("(")[0].replace("void ", ""), "calls": int(r["Calls"]), "total_ms": round(float(r["TotalDurationNs"]) / 1e6, 1),
         "avg_ms": round(float(r["AverageNs"]) / 1e6, 3), "percent": float(r["Percentage"]), "stream": stream_of(r["Name"])} for r in rows[:10]]
    main_rows = [r for r in rows if stream_of(r["Name"]) == "main"]
    main_total = sum(float(r["TotalDurationNs"]) for r in main_rows)
    out["top_main_stream_kernels"] = [
        {"kernel": r["Name"].split("(")[0].replace("void ", ""), "calls": int(r["Calls"]), "avg_ms": round(float(r["AverageNs"]) / 1e6, 3),
         "percent_of_main_stream_time": round(100.0 * float(r["TotalDurationNs"]) / main_total, 2)} for r in main_rows[:10]]
    for r in rows:
        if is_dom(r["Name"]):
            out["rocprof_calls"] = int(r["Calls"])
            out["rocprof_avg_ms"] = float(r["AverageNs"]) / 1e6
            out["rocprof_total_ms"] = float(r["TotalDurationNs"]) / 1e6
            out["rocprof_percentage"] = float(r["Percentage"])
    bench = json.loads((src / "bench.json").read_text().strip().splitlines()[-1])
    out["bench_value_images_per_s"] = bench["value"]
    out["bench_roofline"] = bench["roofline"]
    out["bench_cpu_baseline"] = bench.get("cpu_baseline")
    out["bench_alt"] = bench.get("alt")
    for name in ("alt_kernel_stats.csv", "alt_layers.txt", "f32_layers.txt", "pmc_layers.txt", "zproj_pmc.txt"):
        if (src / name).exists():
            shutil.copy(src / name, dst / f"{rnd}_{name}")

    def counter(tag, cname):
        vals = []
        p = src / f"pmc_{tag}.csv"
        if not p.exists():
            return vals
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] == cname and is_dom(r["Kernel_Name"]):
                vals.append(float(r["Counter_Value"]))
        # the run starts with an 8-patch warm-up forward: keep the full-size (1600-patch) launches only
        return [v for v in vals if v >= 0.25 * max(vals)] if vals else vals

    f = counter("FETCH_SIZE", "FETCH_SIZE")
    w = counter("WRITE_SIZE", "WRITE_SIZE")
    if f and w:
        fetch = 2.0 * sum(f) / len(f) * 1024.0          # gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads
        write = sum(w) / len(w) * 1024.0
        out["pmc_launches"] = len(f)
        out["hbm_read_bytes_per_launch"] = fetch
        out["hbm_write_bytes_per_launch"] = write
        out["traffic_bytes_per_launch"] = fetch + write
        out["traffic_note"] = ("mean over the launches of the dominant kernel in two UNet forwards of 1600 patches (tools/gpu_quick.py: the "
                               "same launches as in the pipeline); FETCH_SIZE x2 per MI355X_MICROARCH.md, WRITE_SIZE as read")
    mf = counter("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES")
    ga = counter("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")
    if mf and ga:
        out["mfma_busy_fraction"] = sum(mf) / (sum(ga) / 8.0 * 1024.0)
    (dst / f"{rnd}_summary.json").write_text(json.dumps(out, indent=2) + "\n")
    print(json.dumps(out, indent=2))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
