R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/obsprof
rm -rf $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools/diag/observables_cost.py > $D.log 2>&1
grep "ms per" $D.log
f=$(find $D -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} total {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
