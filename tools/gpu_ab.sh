#!/bin/bash
# dev tool (GPU box): per-layer timings of the shipped library under a list of environment settings and of every build_variants/ library
# usage: bash tools/gpu_ab.sh <tag> "ENV1=.. ENV2=.." "ENV3=.."   (each quoted argument = one run of tools/gpu_layers.sh)
cd $GRAFT_REPO_ROOT
TAG=$1; shift
i=0
for envs in "" "$@"; do
  echo "== default build, env: [$envs]"
  env $envs timeout -k 10 300 bash tools/gpu_layers.sh ${TAG}_e$i 1600 2>&1 | grep -v "^[EW]2026" | grep "ms"
  i=$((i+1))
done
for so in build_variants/libtmat_*.so; do
  [ -f "$so" ] || continue
  name=$(basename $so .so)
  echo "== $name"
  TMAT_HIP_LIB=$GRAFT_REPO_ROOT/$so timeout -k 10 300 bash tools/gpu_layers.sh ${TAG}_$name 1600 2>&1 | grep "conv_mfma.*64, \|total"
done
