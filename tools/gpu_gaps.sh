#!/bin/bash
# dev tool (GPU box): idle gaps between consecutive kernels of one 1600-patch UNet forward (rocprofv3 --kernel-trace of tools/gpu_quick.py)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/gaps
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/gpu_quick.py 1600 1 > $OUT/run.log 2>&1
T=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$T" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "stem_kernel" in r["Kernel_Name"])
rs = rows[last:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e6
span = (int(rs[-1]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"])) / 1e6
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rs, rs[1:])]
print(f"{len(rs)} kernels: busy {busy:.3f} ms, span {span:.3f} ms, gaps total {sum(gaps)/1e3:.3f} ms, mean {sum(gaps)/len(gaps):.1f} us, max {max(gaps):.1f} us")
PY
rm -rf $OUT/*/
