#!/bin/bash
# dev tool (GPU box): the persistence sweeps on the device against the host threads -- the kernel's duration on an idle chip, then
# bench A/B in one call (same box; shipped library and every build_variants/libtmat_lv*.so), pass trace, kernel trace under the pipeline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 240 python -m pytest tests/test_gpu_dmt.py tests/test_gpu_poison.py -x -q > gpurun_out/dmt_tests.log 2>&1; rc=$?; tail -3 gpurun_out/dmt_tests.log; [ $rc -eq 0 ] || exit 1
export TMPDIR=/tmp
for so in tissue-model-analysis-tools_amd/tmat_amd/libtmat_hip.so build_variants/libtmat_lv*.so; do
  [ -f "$so" ] || continue
  rm -rf gpurun_out/dmt_idle
  (cd /tmp && TMAT_HIP_LIB=$GRAFT_REPO_ROOT/$so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dmt_idle -- python3 $GRAFT_REPO_ROOT/tools/dev/dmt_kernel_time.py > /dev/null 2>&1) || exit 1
  echo "== idle chip, one field per launch, $so"
  find gpurun_out/dmt_idle -name "*kernel_stats.csv" -exec grep "dmt_levels" {} \; | awk -F'",' '{print $2}'
done
for run in "0:" "1:" "0:" "1:"; do
  dev=${run%%:*}; so=${run#*:}
  [ -n "$so" ] && [ ! -f "$so" ] && continue
  echo "== TMAT_DMT_SWEEP_DEVICE=$dev $so"
  if [ -n "$so" ]; then export TMAT_HIP_LIB=$GRAFT_REPO_ROOT/$so; else unset TMAT_HIP_LIB; fi
  TMAT_DMT_SWEEP_DEVICE=$dev TMAT_TRACE=1 timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 --no-alt --no-cpu-baseline > gpurun_out/dmt_ab_$dev.json 2> gpurun_out/dmt_ab_$dev.err || exit 1
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/dmt_ab_$dev.json').read()); print(d['value'], d['unit'], d.get('ms_per_step'))"
  grep "host pass" gpurun_out/dmt_ab_$dev.err | tail -2
done
unset TMAT_HIP_LIB
rm -rf gpurun_out/dmt_trace
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dmt_trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-alt --no-cpu-baseline > /dev/null 2>&1)
find gpurun_out/dmt_trace -name "*kernel_stats.csv" -exec grep "dmt_" {} \; | cut -c1-60,180-300
