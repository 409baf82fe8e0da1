#!/bin/bash
# dev tool (GPU box): cycle accounting of the dominant convolution kernel (build_variants/libtmat_convdiag.so =
# tools/build_variant.sh convdiag "-DTMAT_DIAG" unet_kernels): per chunk and wave the time in the step's work (reads, DMA issue, MFMA issue),
# in the DMA wait and in the barrier; per tile the fill (prologue to first barrier) and the epilogue
cd $GRAFT_REPO_ROOT
TMAT_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/libtmat_convdiag.so timeout -k 10 300 python3 tools/gpu_quick.py 1600 1 2>&1 | grep "convdiag" | tail -3
