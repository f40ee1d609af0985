#!/bin/bash
# What the waves of flow_eval_kernel wait for: issue / wait / fetch counters (one rocprofv3 --pmc pass per group)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_flow_stalls
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_WAIT_IFETCH SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU" "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM"; do
  n=$(echo $c | tr ' ' '+')
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$n -- python3 $R/tools/probe_flow.py C2 > $OUT/$n.log 2>&1 || { echo "  failed $n: $(tail -2 $OUT/$n.log | tr '\n' ' ')" >> $OUT/progress.log; continue; }
  f=$(find $OUT/$n -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $OUT/progress.log <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "flow_eval_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"  {k}: {sum(v)/len(v):.4g}")
PY
done
cat $OUT/progress.log
