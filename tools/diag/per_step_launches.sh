#!/bin/bash
# Round 6: what ONE steady-state headline step launches.  Two kernel traces of bench.py with S and 2 S timed steps; the difference of the per-kernel
# call counts is exactly S steps' worth -- the planning call, the warm-up and the set-up cancel.  Prints calls per step and mean duration of every kernel
# (library GEMMs `Cijk_*`, framework kernels `at::native::*` and copies `__amd_rocclr_copyBuffer` are what VERDICT round 5, item 4 counts).
R=${GRAFT_REPO_ROOT:-$PWD}
S=${1:-40}          # BENCH_ARGS="--solver eigh": the census of another route
OUT=$R/gpurun_out/per_step_launches
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in $S $((2 * S)); do
  timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/s$n -- python3 $R/bench.py --steps $n --warmup 5 --no-cpu-baseline --no-beside $BENCH_ARGS > $OUT/s$n.log 2>&1 || { tail -5 $OUT/s$n.log; exit 1; }
done
python3 - $OUT $S <<'PY'
import csv, glob, sys, os
out, S = sys.argv[1], int(sys.argv[2])
def load(d):
    acc = {}
    for pat in ("*kernel_stats.csv", "*memory_copy_stats.csv"):
        for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
            for r in csv.DictReader(open(f)):
                acc[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]))
    return acc
a, b = load(os.path.join(out, f"s{S}")), load(os.path.join(out, f"s{2 * S}"))
rows = []
for k in b:
    c0, t0 = a.get(k, (0, 0.0))
    c1, t1 = b[k]
    if c1 - c0 > 0:
        rows.append((k, (c1 - c0) / S, (t1 - t0) / max(c1 - c0, 1) / 1e3, (t1 - t0) / S / 1e3))
rows.sort(key=lambda r: -r[3])
print(f"# per steady-state headline step (difference of two traces with {S} and {2 * S} timed steps)")
print(f"{'kernel / copy':100s} {'calls/step':>10s} {'mean us':>9s} {'us/step':>9s}")
for k, c, m, t in rows:
    print(f"{k[:100]:100s} {c:10.2f} {m:9.1f} {t:9.1f}")
lib = sum(c for k, c, m, t in rows if k.startswith("Cijk_") or "rocblas" in k)
fw = sum(c for k, c, m, t in rows if "at::native" in k)
cp = sum(c for k, c, m, t in rows if "copyBuffer" in k or k.startswith("MEMORY_COPY"))
print(f"# library GEMM / BLAS launches per step: {lib:.2f}; framework (at::native) launches per step: {fw:.2f}; copies per step: {cp:.2f}")
PY
