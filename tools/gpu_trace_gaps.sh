#!/bin/bash
# dev tool (GPU box): rocprofv3 kernel trace of one bench step; tools/dev/trace_gaps.py then lists the gaps between the network's kernels
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/gap_trace
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gap_trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-alt --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
T=$(find gpurun_out/gap_trace -name "*kernel_trace.csv" | head -1); python3 tools/dev/trace_gaps.py $T; python3 tools/dev/trace_overlap.py $T
rm -rf gpurun_out/gap_trace
