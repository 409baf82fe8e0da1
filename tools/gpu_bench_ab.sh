#!/bin/bash
# dev tool (GPU box): bench.py under alternating environment settings in one call (same box: the only valid comparison on this pool)
# usage: bash tools/gpu_bench_ab.sh <repetitions> "ENV_A=.." "ENV_B=.." ...   (an empty string = the defaults)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
REP=$1; shift
for r in $(seq 1 $REP); do
  for envs in "$@"; do
    echo "== [$envs]"
    env $envs TMAT_TRACE=0 timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 --no-alt --no-cpu-baseline > gpurun_out/bench_ab.json 2> gpurun_out/bench_ab.err || exit 1
    python3 -c "import json; d=json.loads(open('gpurun_out/bench_ab.json').read()); print(d['value'], d['unit'], d['ms_per_step'], 'dominant', d['roofline']['avg_launch_ms'], 'path', d['roofline']['path_frac'])"
  done
done
