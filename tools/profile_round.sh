#!/bin/bash
# Round profile bundle (run on the GPU box through gpurun): default bench line, rocprofv3 --kernel-trace --stats of
# the same command, and separate PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy) on a short run of the same pipeline.
# Usage: bash tools/profile_round.sh r01 [no-pmc | pmc-only]      (two calls when one does not fit the gpurun time limit)
set -e
R=${1:-r01}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
cd $ROOT
if [ "$2" != "pmc-only" ]; then
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-alt > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/trace            # the raw per-dispatch trace is large; the stats summary is what is kept
echo "kernel-trace done"
fi
if [ "$2" = "no-pmc" ]; then ls -la $OUT; exit 0; fi
# PMC passes (separate runs, --kernel-trace only beside --pmc) on the UNet forward alone: the same kernels with the same
# 1600-patch launches as in the pipeline (tools/gpu_quick.py), without the many short morphology / thinning launches,
# each of which costs milliseconds under counter collection
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  tag=$(echo $grp | cut -d' ' -f1)
  echo "pmc $tag ..."
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$tag -- python3 $ROOT/tools/gpu_quick.py 1600 2 > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.err || echo "pmc $tag failed"
  find $OUT/pmc_$tag -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $OUT/pmc_$tag.csv || true
  find $OUT/pmc_$tag -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $OUT/pmc_${tag}_trace.csv || true
  rm -rf $OUT/pmc_$tag
  echo "pmc $tag done"
done
# per-layer durations of one 1600-patch forward, f32 and the opt-in bf16x3 path (rocprofv3 --kernel-trace of tools/gpu_quick.py)
cd $ROOT
bash tools/gpu_layers.sh ${R}_f32 1600 > /dev/null 2>&1 && cp gpurun_out/layers_${R}_f32/layers.txt $OUT/f32_layers.txt
TMAT_PRECISION=bf16x3 bash tools/gpu_layers.sh ${R}_alt 1600 > /dev/null 2>&1 && cp gpurun_out/layers_${R}_alt/layers.txt $OUT/alt_layers.txt
cd /tmp
TMAT_PRECISION=bf16x3 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/alt_trace -- python3 $ROOT/tools/gpu_quick.py 1600 2 > $OUT/alt_quick.log 2>&1 || echo "alt trace failed"
find $OUT/alt_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/alt_kernel_stats.csv || true
rm -rf $OUT/alt_trace
cd $ROOT
bash tools/gpu_pmc_layers.sh $R 1600 kernel > $OUT/pmc_layers.txt 2>&1 || echo "pmc layers failed"
ls -la $OUT
