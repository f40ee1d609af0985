set -e
for v in 1 0; do
  export VMCPDE_PIVOT_HINTS=$v
  for sv in cholesky matrixfree eigh eigh-matrixfree; do
    python bench.py --steps 30 --warmup 6 --no-cpu-baseline --solver $sv > gpurun_out/h${v}_$sv.json 2> gpurun_out/h${v}_$sv.err
    python -c "
import json
l=json.load(open('gpurun_out/h${v}_$sv.json'))
print('hints $v $sv', round(l['value'],2), 'steps/s', round(l['ms_per_step'],3), 'ms', l['rank_stage_ms'][0], (l['config'].get('cholesky_engine') or l['config'].get('spectral')), flush=True)"
  done
done
