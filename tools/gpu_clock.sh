#!/bin/bash
# dev tool (GPU box): average shader clock and MFMA busy per conv launch for each build_variants/libtmat_*.so
# (GRBM_GUI_ACTIVE / kernel duration; separate PMC run with --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
for so in $GRAFT_REPO_ROOT/build_variants/libtmat_*.so; do
  name=$(basename $so .so)
  OUT=$GRAFT_REPO_ROOT/gpurun_out/clock_$name
  mkdir -p $OUT
  export TMAT_HIP_LIB=$so
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/gpu_quick.py ${1:-1600} 1 > $OUT/run.log 2>&1
  C=$(find $OUT -name "*counter_collection.csv" | head -1)
  T=$(find $OUT -name "*kernel_trace.csv" | head -1)
  echo "== $name"
  python3 - "$C" "$T" <<'PY'
import csv, sys, collections
cnt = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    cnt[r["Dispatch_Id"]]["name"] = r["Kernel_Name"]
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rows = [(int(k), v) for k, v in cnt.items() if "conv_mfma" in v["name"] and (", 3, " in v["name"] or ", 2, " in v["name"])]
rows.sort()
for k, v in rows[-8:]:
    d = dur.get(str(k))
    g = v.get("GRBM_GUI_ACTIVE", 0)
    print(f"{d/1e3:8.3f} ms  clock {g/d/1e3 if d else 0:6.3f} GHz  mfma_busy/gui {v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/g if g else 0:7.3f}  sq_busy/gui {v.get('SQ_BUSY_CYCLES',0)/g if g else 0:7.3f}  {v['name'][:70]}")
PY
  rm -f $T $C
done
