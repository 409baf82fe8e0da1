#!/bin/bash
# rocprofv3 kernel trace of the Z-stack branch (tools/bench_stack.py); summaries land in gpurun_out/stack_prof/
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/stack_prof
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 tools/bench_stack.py --steps 3 --no-cpu > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_stack.py --steps 2 --no-cpu > $OUT/bench_prof.json 2> $OUT/prof.err
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -30 $OUT/kernel_stats.csv | cut -c1-200
