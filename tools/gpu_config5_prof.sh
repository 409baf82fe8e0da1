#!/bin/bash
# bench + rocprofv3 kernel trace of config #5 (tools/bench_config5.py); summaries land in gpurun_out/config5_prof/
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/config5_prof
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 tools/bench_config5.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_config5.py --steps 1 --no-cpu > $OUT/bench_prof.json 2> $OUT/prof.err
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -16 $OUT/kernel_stats.csv | cut -c1-190
