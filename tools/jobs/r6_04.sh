set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_04
mkdir -p $O
python -m pytest tests/test_gpu_lowrank_eigh.py tests/test_gpu_headline_parity.py tests/test_gpu_trajectory_attribution.py tests/test_gpu_solve.py tests/test_gpu_tdvp.py tests/test_gpu_multirank.py -x -q -m gpu --durations=6 > $O/t1.log 2>&1 || { tail -60 $O/t1.log; exit 1; }
tail -12 $O/t1.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-beside > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
VMCPDE_FROZEN_PIVOTS=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-beside > $O/bench_nofrozen.json 2> $O/bench_nofrozen.err || { tail -20 $O/bench_nofrozen.err; exit 1; }
python - <<'PY'
import json
for f in ('bench','bench_nofrozen'):
    d=json.loads(open(f'gpurun_out/r6_04/{f}.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['rank_stage_ms'], d['config'].get('solver_residual'))
PY
timeout -k 10 300 python tools/run_headline_trajectory.py > $O/headline_traj_f64.txt 2>&1 || { tail -20 $O/headline_traj_f64.txt; exit 1; }
cat $O/headline_traj_f64.txt
