#!/bin/bash
# rocprofv3 kernel-trace summary of a short bench.py run (whole pipeline); run on the GPU box
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/profb_$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --images ${2:-16} --distinct 4 --warmup 1 --steps 1 --no-cpu-baseline > $OUT/run.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
cat $OUT/kernel_stats.csv | cut -c1-150
tail -2 $OUT/run.log | cut -c1-400
