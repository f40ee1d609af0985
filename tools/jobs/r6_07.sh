set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_07
mkdir -p $O
python -m pytest tests/test_gpu_lowrank_eigh.py tests/test_gpu_headline_parity.py -x -q -m gpu > $O/t1.log 2>&1 || { tail -60 $O/t1.log; exit 1; }
tail -3 $O/t1.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-beside > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_07/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['rank_stage_ms'])
PY
