#!/bin/bash
# dev tool (GPU box): per-layer conv timings for every build_variants/libtmat_*.so
cd $GRAFT_REPO_ROOT
for so in build_variants/libtmat_*.so; do
  name=$(basename $so .so)
  echo "== $name"
  TMAT_HIP_LIB=$GRAFT_REPO_ROOT/$so timeout -k 10 200 bash tools/gpu_layers.sh $name ${1:-1600} 2>&1 | grep "conv_mfma.*, [23], \|total"
done
