#!/bin/bash
# dev tool (GPU box): FETCH_SIZE / duration per dispatch of the 3x3 and sub-pixel convolutions for a library variant ($1 = .so path or "" for the shipped one)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/fetch_$2
mkdir -p $OUT
[ -n "$1" ] && export TMAT_HIP_LIB=$1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/run -- python3 $GRAFT_REPO_ROOT/tools/gpu_quick.py 1600 2 > $OUT/log.txt 2>&1
C=$(find $OUT/run -name "*counter_collection.csv" | head -1)
python3 - "$C" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE" and "conv_mfma_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 5000000 and (", 3, " in r["Kernel_Name"] or ", 2, " in r["Kernel_Name"]):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        print(f"{float(r['Counter_Value']) * 1024 / 1e9:8.2f} GB raw  {d:7.2f} ms  grid {r['Grid_Size']:>10}  {r['Kernel_Name'].split('(')[0][-45:]}")
PY
rm -rf $OUT/run
