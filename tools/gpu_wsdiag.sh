#!/bin/bash
# dev tool (GPU box): phase stamps of the wave-specialised separable kernel (build_variants/libtmat_wsdiag.so)
cd $GRAFT_REPO_ROOT
TMAT_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/libtmat_wsdiag.so timeout -k 10 300 python3 tools/gpu_quick.py 1600 1 2>&1 | grep "wsdiag\|wstl\|rep" | tail -104
