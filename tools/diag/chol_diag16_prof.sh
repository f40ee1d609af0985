R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for m in 1 0; do
  export VMCPDE_CHOL_DIAG16=$m
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/diag16prof_$m -- python3 $R/tools/probe_cholesky.py 640 > $R/gpurun_out/diag16prof_$m.log 2>&1
  f=$(find $R/gpurun_out/diag16prof_$m -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $m <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'chol_' in r['Name']: print("diag16 =",sys.argv[2], r['Name'][23:60], r['Calls'], 'avg', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1))
PY
done
