for ns in 0 5 6 7 8 9 10 12 14; do
  if [ $ns = 0 ]; then unset VMCPDE_GRAM_NSPLIT; else export VMCPDE_GRAM_NSPLIT=$ns; fi
  python bench.py --workload C4 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/c4_ns$ns.json 2>gpurun_out/c4_ns$ns.err || exit 1
  python -c "
import json,sys
l=json.load(open('gpurun_out/c4_ns$ns.json'))
print('nsplit $ns', round(l['ms_per_step'],3), 'gram', round(l['roofline']['ms_per_launch'],3), 'frac', round(l['roofline']['frac'],4), flush=True)"
done
