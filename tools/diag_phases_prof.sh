#!/bin/bash
# chol_diag_kernel / chol_syrk_diag_kernel durations per cumulative diag phase (rocprofv3 kernel stats; small matrix so the
# diagonal block is the only long workgroup of the fused launch)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/diagprof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ph in 0 1 3 4 5; do
  export VMCPDE_DIAG_PHASE=$ph
  echo "[$(date +%T)] phase $ph" >> $OUT/progress.log
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ph$ph -- python3 $R/tools/probe_cholesky.py 640 > $OUT/ph$ph.log 2>&1 || { echo "phase $ph failed" >> $OUT/progress.log; exit 1; }
  f=$(find $OUT/ph$ph -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $ph >> $OUT/progress.log <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'chol_' in r['Name']: print("  phase",sys.argv[2], r['Name'][23:50], r['Calls'], round(float(r['AverageNs'])/1e3,1), round(float(r['MinNs'])/1e3,1))
PY
done
