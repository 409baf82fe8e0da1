#!/bin/bash
# PMC passes on the UNet patch batch (dev tool); separate runs per counter group (MI355X_MICROARCH.md: rocprofv3 PMC slots)
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
mkdir -p $OUT
for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/tools/gpu_quick.py 200 1 > $OUT/$tag.log 2>&1 || echo "pmc group $tag failed"
  find $OUT/$tag -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $OUT/$tag.csv || true
done
ls $OUT
