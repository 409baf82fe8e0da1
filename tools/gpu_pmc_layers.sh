#!/bin/bash
# dev tool (GPU box): per-dispatch PMC figures of one UNet forward over a full pass: shader clock, MFMA busy, LDS conflicts,
# wait buckets.  Separate rocprofv3 --pmc runs per counter group, each with --kernel-trace only.
# Normalisation (stated here because the raw sums are not fractions): GRBM_GUI_ACTIVE is summed over the 8 XCDs, so the
# shader clock is GUI / 8 / duration; SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, so the matrix-pipe busy
# fraction is MFMA_BUSY / (GUI / 8 x 1024); the SQ_WAIT_* / SQ_ACTIVE_* buckets are fractions of SQ_WAVE_CYCLES.
# Usage: bash tools/gpu_pmc_layers.sh <tag> [patches] [kernel-name filter]
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcl_$1
mkdir -p $OUT
N=${2:-1600}
F=${3:-mfma}
i=0
for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 $GRAFT_REPO_ROOT/tools/gpu_quick.py $N 1 > $OUT/g$i.log 2>&1 || echo "pmc group $i failed"
  C=$(find $OUT/g$i -name "*counter_collection.csv" | head -1)
  T=$(find $OUT/g$i -name "*kernel_trace.csv" | head -1)
  python3 - "$C" "$T" "$F" <<'PY' | tee $OUT/g$i.txt
import csv, sys, collections
cnt = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    cnt[r["Dispatch_Id"]]["name"] = r["Kernel_Name"]
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rows = sorted((int(k), v) for k, v in cnt.items())
last = max(k for k, v in rows if "stem_kernel" in v["name"] or "stem_even_kernel" in v["name"])
for k, v in rows:
    if k < last or sys.argv[3] not in v["name"]:
        continue
    d = dur.get(str(k), 0.0)
    name = v["name"].replace("tmat::", "").replace("void ", "").split("(")[0][:44]
    keys = [c for c in v if c != "name"]
    if "GRBM_GUI_ACTIVE" in v:
        # rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs and SQ_VALU_MFMA_BUSY_CYCLES summed over the 1024 SIMDs
        # (256 CUs x 4): clock = GUI / 8 / duration; matrix-pipe busy fraction = MFMA_BUSY / (GUI / 8 * 1024)
        g = v["GRBM_GUI_ACTIVE"] / 8.0
        print(f"{d/1e3:8.3f} ms clock {g/d/1e3 if d else 0:5.2f} GHz mfma_busy {v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(g*1024.0) if g else 0:6.3f}  {name}")
    else:
        wc = v.get("SQ_WAVE_CYCLES", 1) or 1
        print(f"{d/1e3:8.3f} ms wait_any {v.get('SQ_WAIT_ANY',0)/wc:5.2f} wait_inst {v.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} active {v.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f} wait_lds {v.get('SQ_WAIT_INST_LDS',0)/wc:5.2f} "
              f"lds_conf/idx {v.get('SQ_LDS_BANK_CONFLICT',0)/(v.get('SQ_LDS_IDX_ACTIVE',1) or 1):5.3f} valu_insts {v.get('SQ_INSTS_VALU',0):.3g}  {name}")
PY
  rm -rf $OUT/g$i
done
