#!/bin/bash
# rocprofv3 kernel-trace summary of the UNet patch batch (dev tool); run on the GPU box
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/gpu_quick.py 200 2 > $OUT/run.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -20 $OUT/kernel_stats.csv
