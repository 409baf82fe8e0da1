#!/bin/bash
# GPU box: the side workloads' numbers for the round bundle -- config #5 (cell area + invasion depth) bench + kernel stats, the
# Z-stack branch bench + kernel stats, and PMC counters of zproj_focus_kernel (is it VALU-bound?).  Output: gpurun_out/side_<tag>/
R=${1:-r03}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/side_$R
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 bash tools/gpu_config5_prof.sh > $OUT/config5.log 2>&1
cp gpurun_out/config5_prof/bench.json $OUT/config5_bench.json 2>/dev/null
cp gpurun_out/config5_prof/kernel_stats.csv $OUT/config5_kernel_stats.csv 2>/dev/null
timeout -k 10 600 bash tools/gpu_stack_prof.sh > $OUT/stack.log 2>&1
cp gpurun_out/stack_prof/bench.json $OUT/stack_bench.json 2>/dev/null
cp gpurun_out/stack_prof/kernel_stats.csv $OUT/stack_kernel_stats.csv 2>/dev/null
timeout -k 10 300 python3 tools/bench_zproj.py --stacks 32 --steps 5 > $OUT/zproj_bench.json 2> $OUT/zproj.err
# BASELINE config #3 end to end: focus stacking, then the 2-D branch on the 2048 x 2048 projections (first line of the output)
timeout -k 10 400 python3 tools/bench_zproj.py --chain --stacks 8 --steps 1 --warmup 1 2> $OUT/chain.err | head -1 > $OUT/config3_chain.json
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/zp_$tag -- python3 $ROOT/tools/bench_zproj.py --stacks 8 --steps 2 --warmup 1 > /dev/null 2> $OUT/zp_$tag.err || echo "pmc $tag failed"
  C=$(find $OUT/zp_$tag -name "*counter_collection.csv" | head -1)
  T=$(find $OUT/zp_$tag -name "*kernel_trace.csv" | head -1)
  python3 - "$C" "$T" >> $OUT/zproj_pmc.txt <<'PY'
import csv, sys, collections
cnt = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    if "zproj_focus" in r["Kernel_Name"]:
        cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(sys.argv[2])) if "zproj_focus" in r["Kernel_Name"]}
for k, v in sorted(cnt.items(), key=lambda kv: int(kv[0]))[-1:]:
    print(f"zproj_focus_kernel, 16 x 2048 x 2048 u16 stacks (one stack per launch), last launch of {dur.get(k, 0):.1f} us: " + ", ".join(f"{c} {x:.6g}" for c, x in sorted(v.items())))
PY
  rm -rf $OUT/zp_$tag
done
cat $OUT/zproj_pmc.txt
head -c 600 $OUT/config5_bench.json; echo; head -c 400 $OUT/stack_bench.json; echo; head -c 500 $OUT/zproj_bench.json
