#!/bin/bash
# dev tool (GPU box): DMT goldens through the device path, then the Z-stack bench with the one-workgroup and the multi-workgroup sort
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_dmt.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -3
for one in 1 0; do
  echo "== TMAT_DMT_SORT_ONE_WG=$one"
  TMAT_DMT_SORT_ONE_WG=$one timeout -k 10 300 python3 tools/bench_stack.py --steps 5 --no-cpu 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d.get('ms_per_step'))"
done
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/dmt_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dmt_trace -- python3 $GRAFT_REPO_ROOT/tools/bench_stack.py --steps 3 --no-cpu > /dev/null 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/dmt_trace -name "*kernel_stats.csv" -exec grep "ms_\|dmt_" {} \; | cut -c1-160
