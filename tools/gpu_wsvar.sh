#!/bin/bash
# dev tool (GPU box): per-layer timings of the separable kernels for every build_variants/libtmat_ws*.so
cd $GRAFT_REPO_ROOT
for so in build_variants/libtmat_ws*.so; do
  name=$(basename $so .so)
  echo "== $name"
  TMAT_HIP_LIB=$GRAFT_REPO_ROOT/$so timeout -k 10 200 bash tools/gpu_layers.sh $name 1600 2>&1 | grep "sepconv\|total"
done
