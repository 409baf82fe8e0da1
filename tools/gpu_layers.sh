#!/bin/bash
# per-dispatch durations of one UNet forward over a full 1600-patch pass (dev tool); run on the GPU box
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/layers_$1
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/gpu_quick.py ${2:-1600} 1 > $OUT/run.log 2>&1
T=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$T" > $OUT/layers.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last forward = everything after the last stem_kernel dispatch
last = max(i for i, r in enumerate(rows) if "stem_kernel" in r["Kernel_Name"] or "stem_even_kernel" in r["Kernel_Name"])
tot = 0
for r in rows[last:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    tot += d
    name = r["Kernel_Name"].replace("tmat::", "").split("(")[0][:60]
    print(f"{d:9.3f} ms  grid {r['Grid_Size_X']:>10} x{r.get('Grid_Size_Y','1'):>2}  {name}")
print(f"{tot:9.3f} ms total")
PY
cat $OUT/layers.txt; tail -3 $OUT/run.log
rm -f $T
