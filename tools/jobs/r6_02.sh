set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_02
mkdir -p $O
python -m pytest tests/test_gpu_lowrank_eigh.py tests/test_gpu_trajectory_attribution.py tests/test_gpu_headline_parity.py tests/test_gpu_tdvp.py tests/test_gpu_solve.py -x -q -m gpu --durations=8 > $O/t1.log 2>&1 || { tail -40 $O/t1.log; exit 1; }
tail -15 $O/t1.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_02/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['rank_stage_ms'], d['roofline']['frac'])
print(json.dumps({k:d['cpu_baseline'][k] for k in ('value','cores','host_cpu_count','affinity_count','thread_sweep')}))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-beside > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err || { tail -20 $GRAFT_REPO_ROOT/$O/prof.err; exit 1; }
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r6_02/kernel_stats.csv')))
for r in rows:
    n=r['Name']
    if 'at::native' in n or 'Cijk' in n or 'copyBuffer' in n or 'fillBuffer' in n or 'woodbury' in n or 'gather' in n or 'rocblas' in n:
        print(n[:120], r['Calls'], r['AverageNs'])
PY
