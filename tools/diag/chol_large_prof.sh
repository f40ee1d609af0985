# Where the time of the large-P Cholesky goes: per-kernel totals and the rate of the trailing updates by size.
#   usage (GPU box): bash tools/diag/chol_large_prof.sh [P]
R=${GRAFT_REPO_ROOT:-$PWD}
P=${1:-16181}
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/chol_large_$P
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools/probe_cholesky.py $P > $D.log 2>&1
tail -4 $D.log
python3 - $D <<'PY'
import csv, sys, glob, collections
d = sys.argv[1]
st = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(st)):
    if float(r['Percentage']) > 0.5:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} total {float(r['TotalDurationNs'])/1e6:9.3f} ms avg {float(r['AverageNs'])/1e3:9.1f} us")
tr = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(tr)) if 'chol_syrk' in r['Kernel_Name'] or 'chol_update' in r['Kernel_Name']]
per = len(rows) // 4
last = rows[-per:]
print("update launches of the last factorisation (workgroups; duration)")
big = sorted(last, key=lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp']))[-16:]
for r in sorted(big, key=lambda r: int(r['Start_Timestamp'])):
    wg = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']) if 'Grid_Size_X' in r else int(r['Grid_Size']) // int(r['Workgroup_Size'])
    us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    print(f"  {r['Kernel_Name'][:40]:40s} workgroups {wg:6d}  {us:8.1f} us")
t0 = int(last[0]['Start_Timestamp']); t1 = int(last[-1]['End_Timestamp'])
print("span of the updates", (t1 - t0) / 1e6, "ms; sum of their durations", sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in last) / 1e6, "ms")
PY
