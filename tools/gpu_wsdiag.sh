#!/bin/bash
# dev tool (GPU box): phase stamps of the wave-specialised separable kernel for every build_variants/libtmat_wsd*.so
cd $GRAFT_REPO_ROOT
for lib in build_variants/libtmat_wsd*.so; do
    echo "== $lib"
    TMAT_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 300 python3 tools/gpu_quick.py 1600 1 2>&1 | grep "wsdiag\|wstl\|rep" | tail -${WSD_TAIL:-104} || exit 1
done
