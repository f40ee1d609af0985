set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_09
mkdir -p $O
python -m pytest tests/test_gpu_flow.py tests/test_gpu_golden.py tests/test_gpu_tdvp.py tests/test_gpu_large_configs.py tests/test_gpu_observables.py tests/test_gpu_trajectory_attribution.py -x -q -m gpu > $O/t1.log 2>&1 || { tail -60 $O/t1.log; exit 1; }
tail -3 $O/t1.log
for v in 1 0; do
  echo "== VMCPDE_FLOW_FAST=$v" >> $O/flow_fast_ab.txt
  for wl in C2 C4; do VMCPDE_FLOW_FAST=$v KEEP_ACT=0 python tools/probe_flow.py $wl 2>&1 | tail -2 >> $O/flow_fast_ab.txt; done
done
cat $O/flow_fast_ab.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-beside > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_09/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['rank_stage_ms'])
PY
