#!/bin/bash
# dev tool (GPU box): bench.py with the shipped library and with every build_variants/libtmat_*.so, alternating, in one call
# usage: bash tools/gpu_lib_ab.sh <repetitions>
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in $(seq 1 ${1:-2}); do
  for so in "" build_variants/libtmat_*.so; do
    [ -n "$so" ] && [ ! -f "$so" ] && continue
    echo "== [${so:-shipped}]"
    if [ -n "$so" ]; then export TMAT_HIP_LIB=$GRAFT_REPO_ROOT/$so; else unset TMAT_HIP_LIB; fi
    timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 --no-alt --no-cpu-baseline > gpurun_out/bench_ab.json 2> gpurun_out/bench_ab.err || exit 1
    python3 -c "import json; d=json.loads(open('gpurun_out/bench_ab.json').read()); print(d['value'], d['unit'], d['ms_per_step'], 'dominant', d['roofline']['avg_launch_ms'], 'path', d['roofline']['path_frac'])"
  done
done
