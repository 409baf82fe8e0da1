#!/bin/bash
# dev tool (GPU box): bench.py for several pass sizes (patches resident per pass = 200 x images per pass) in one call
# usage: bash tools/gpu_bench_mp.sh <repetitions> 1600 1800 ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
REP=$1; shift
for r in $(seq 1 $REP); do
  for mp in "$@"; do
    echo "== [max_patches $mp]"
    TMAT_TRACE=0 timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 --no-alt --no-cpu-baseline --max-patches $mp > gpurun_out/bench_mp.json 2> gpurun_out/bench_mp.err || exit 1
    python3 -c "import json; d=json.loads(open('gpurun_out/bench_mp.json').read()); print(d['value'], d['unit'], d['ms_per_step'], 'dominant', d['roofline']['avg_launch_ms'], 'path', d['roofline']['path_frac'])"
  done
done
