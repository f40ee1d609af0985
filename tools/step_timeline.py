"""Timeline of one TDVP step from a rocprofv3 --kernel-trace csv: per kernel family the busy time and the idle gaps in front
of it, for the step between the last two gram kernels of the trace.

    python tools/step_timeline.py <..._kernel_trace.csv> [--list]
"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:60]
grams = [i for i, r in enumerate(rows) if "gram_dma_kernel" in r["Kernel_Name"]]
# a step = from the first kernel after gram[-3]'s solve ... simpler: window between the starts of the last two Gram launches
k = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else 5      # default: inside bench.py's un-synchronised timed loop
lo, hi = grams[k], grams[k + 1]
win = rows[lo:hi]
t0 = int(win[0]["Start_Timestamp"])
busy, gaps, calls = defaultdict(float), defaultdict(float), defaultdict(int)
prev_end = t0
for r in win:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = name(r)
    busy[n] += (e - s) / 1e3
    gaps[n] += max(0, s - prev_end) / 1e3
    calls[n] += 1
    if "--list" in sys.argv:
        print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f}  gap {max(0, s - prev_end) / 1e3:6.1f}  {n}")
    prev_end = max(prev_end, e)
total = (int(rows[hi]["Start_Timestamp"]) - t0) / 1e3
print(f"step window {total:.1f} us, {len(win)} launches")
for n in sorted(busy, key=lambda k: -(busy[k] + gaps[k])):
    print(f"{n:60s} calls {calls[n]:4d}  busy {busy[n]:9.1f} us  gaps-before {gaps[n]:8.1f} us")
print(f"sum busy {sum(busy.values()):.1f}  sum gaps {sum(gaps.values()):.1f}")
