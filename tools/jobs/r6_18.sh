set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_18
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1 || { tail -60 $O/t_all.log; exit 1; }
tail -3 $O/t_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -2 $O/smoke.log
python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python bench.py --solver eigh --steps 10 --warmup 3 --no-cpu-baseline --no-beside > $O/bench_eigh.json 2> $O/bench_eigh.err || { tail -20 $O/bench_eigh.err; exit 1; }
python bench.py --workload C3 --solver eigh --steps 5 --warmup 2 --no-cpu-baseline --no-beside > $O/bench_C3_eigh.json 2> $O/bench_C3_eigh.err || { tail -20 $O/bench_C3_eigh.err; exit 1; }
python bench.py --solver cholesky-dense --steps 10 --warmup 3 --no-cpu-baseline --no-beside > $O/bench_cholesky_dense.json 2> $O/bench_cholesky_dense.err || { tail -20 $O/bench_cholesky_dense.err; exit 1; }
python bench.py --workload C4 --solver eigh --steps 5 --warmup 2 --no-cpu-baseline --no-beside > $O/bench_C4_eigh.json 2> $O/bench_C4_eigh.err || { tail -20 $O/bench_C4_eigh.err; exit 1; }
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6_18/bench*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], round(d['value'],3), round(d['ms_per_step'],3), d['rank_stage_ms'], round(d['roofline']['frac'],4))
PY
