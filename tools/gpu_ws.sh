#!/bin/bash
# dev tool (GPU box): parity of the UNet kernels, then per-layer timings with the wave-specialised and the round-2 separable kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_unet.py -x -q > gpurun_out/ws_tests.log 2>&1
rc=$?
tail -5 gpurun_out/ws_tests.log
[ $rc -ne 0 ] && exit $rc
TMAT_SEP_WS=1 timeout -k 10 300 bash tools/gpu_layers.sh ws1 1600 > gpurun_out/ws1.log 2>&1 && grep "sepconv\|total\|pool_fix" gpurun_out/ws1.log
TMAT_SEP_WS=0 timeout -k 10 300 bash tools/gpu_layers.sh ws0 1600 > gpurun_out/ws0.log 2>&1 && grep "sepconv\|total\|pool_fix" gpurun_out/ws0.log
