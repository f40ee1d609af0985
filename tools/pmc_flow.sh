#!/bin/bash
# PMC counters of flow_eval_kernel (one pass per group): LDS bank conflicts, VALU/LDS/SMEM instruction mix, busy cycles
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_flow
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  n=$(echo $c | tr ' ' '+')
  echo "[$(date +%T)] $n" >> $OUT/progress.log
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$n -- python3 $R/tools/probe_flow.py C2 > $OUT/$n.log 2>&1 || { echo "  failed: $(tail -2 $OUT/$n.log | tr '\n' ' ')" >> $OUT/progress.log; continue; }
  f=$(find $OUT/$n -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $OUT/progress.log <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "flow_eval_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"  {k}: {sum(v)/len(v):.4g} per launch ({len(v)} launches)")
PY
done
cat $OUT/progress.log
