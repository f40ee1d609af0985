set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_03
mkdir -p $O
python -m pytest tests/test_gpu_flow.py tests/test_gpu_golden.py tests/test_gpu_large_configs.py -x -q -m gpu > $O/t_flow.log 2>&1 || { tail -40 $O/t_flow.log; exit 1; }
tail -3 $O/t_flow.log
for wl in C2 C4 C3; do
  for v in "" fullbarrier; do
    if [ -n "$v" ]; then export VMCPDE_LIB=$PWD/vmc_pde_amd/_lib/libvmcpde_$v.so; else unset VMCPDE_LIB; fi
    echo "== $wl variant '$v'" >> $O/flow_barrier_ab.txt; KEEP_ACT=0 python tools/probe_flow.py $wl 2>&1 | tail -2 >> $O/flow_barrier_ab.txt
  done
done
unset VMCPDE_LIB
cat $O/flow_barrier_ab.txt
timeout -k 10 300 python tools/diag/flow_dephase_probe.py C2 > $O/flow_dephase_C2.txt 2>&1 || { tail -20 $O/flow_dephase_C2.txt; exit 1; }
cat $O/flow_dephase_C2.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-beside > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err || { tail -20 $GRAFT_REPO_ROOT/$O/prof.err; exit 1; }
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r6_03/kernel_stats.csv')))
for r in rows:
    n=r['Name']
    if any(k in n for k in ('at::native','Cijk','copyBuffer','fillBuffer','woodbury','gather','rocblas','flow_eval','gram_dma')):
        print(n[:130], r['Calls'], r['AverageNs'])
PY
