set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_13
mkdir -p $O
python -m pytest tests/test_gpu_flow.py -x -q -m gpu -k "from_latent or fused" > $O/t0.log 2>&1 || { tail -60 $O/t0.log; exit 1; }
tail -3 $O/t0.log
python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1 || { tail -60 $O/t_all.log; exit 1; }
tail -3 $O/t_all.log
for v in 1 0; do
  VMCPDE_FUSED_SAMPLE_EVAL=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-beside > $O/bench_fused$v.json 2> $O/bench_fused$v.err || { tail -20 $O/bench_fused$v.err; exit 1; }
  VMCPDE_FUSED_SAMPLE_EVAL=$v python bench.py --workload C4 --steps 20 --warmup 5 --no-cpu-baseline --no-beside > $O/bench_C4_fused$v.json 2> $O/bench_C4_fused$v.err || { tail -20 $O/bench_C4_fused$v.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6_13/bench*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], round(d['value'],3), round(d['ms_per_step'],3), d['rank_stage_ms'])
PY
