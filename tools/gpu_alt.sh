#!/bin/bash
# dev tool (GPU box): the split-precision tests, then per-layer timings of the bf16x3 path
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_alt_precision.py -x -q -s > gpurun_out/alt_tests.log 2>&1
rc=$?
grep "bf16x\|random weights\|passed\|failed\|Error\|max |pred" gpurun_out/alt_tests.log | head -20
[ $rc -ne 0 ] && exit $rc
for m in bf16x3 bf16x6; do echo "== $m"; TMAT_PRECISION=$m timeout -k 10 300 bash tools/gpu_layers.sh alt_$m 1600 > gpurun_out/alt_layers_$m.log 2>&1 && grep "conv_mfma.*, 3, \|total" gpurun_out/alt_layers_$m.log; done
