# kernel busy time against wall time of a reference-sized TDVP call (P = 411, N = 10 000, Cholesky path): how launch-bound is it?
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/small_timeline
rm -rf $D
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/tools/probe_small.py > $D.log 2>&1
grep "cholesky" $D.log
f=$(find $D -name "*kernel_trace.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# the cholesky-path loop comes first in probe_small.py: take launches between the 10th and 50th chol_diag16_kernel
idx = [i for i, r in enumerate(rows) if "chol_diag16_kernel" in r["Kernel_Name"] or "chol_update_diag16" in r["Kernel_Name"] and False]
first = [i for i, r in enumerate(rows) if "chol_diag16_kernel" in r["Kernel_Name"]]
a, b = first[10], first[50]
win = rows[a:b]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in win)
wall = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
print(f"40 calls: {len(win) / 40:.1f} launches per call, wall {wall / 40 / 1e3:.1f} us per call, kernels busy {busy / 40 / 1e3:.1f} us per call ({100.0 * busy / wall:.0f} %)")
from collections import defaultdict
agg = defaultdict(lambda: [0, 0])
for r in win:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:50]
    agg[n][0] += 1; agg[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {n:50s} {c / 40:5.1f} per call {t / 40 / 1e3:8.1f} us")
PY
