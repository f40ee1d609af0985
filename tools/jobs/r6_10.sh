set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_10
mkdir -p $O
python -m pytest tests/test_gpu_gram.py tests/test_gpu_large_configs.py tests/test_gpu_headline_parity.py tests/test_gpu_multirank.py -x -q -m gpu > $O/t1.log 2>&1 || { tail -60 $O/t1.log; exit 1; }
tail -3 $O/t1.log
for wl in C4 C2; do
  python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-beside > $O/bench_$wl.json 2> $O/bench_$wl.err || { tail -20 $O/bench_$wl.err; exit 1; }
done
python - <<'PY'
import json
for wl in ('C4','C2'):
    d=json.loads(open(f'gpurun_out/r6_10/bench_{wl}.json').read().strip().splitlines()[-1])
    print(wl, d['value'], d['ms_per_step'], d['rank_stage_ms'], d['roofline']['frac'])
PY
