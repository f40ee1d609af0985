cd /tmp && export TMPDIR=/tmp
for v in "" gold; do
  D=$GRAFT_REPO_ROOT/gpurun_out/chol_trace_$v; rm -rf $D
  if [ -n "$v" ]; then export VMCPDE_LIB=$GRAFT_REPO_ROOT/vmc_pde_amd/_lib/libvmcpde_gold.so; fi
  rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/tools/probe_cholesky.py 4213 > $D.log 2>&1
  f=$(find $D -name "*kernel_trace.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'chol' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last factorisation: last 97 launches (1 diag + 32*(2 strips + 1 update))
last=rows[-97:]
t0=int(last[0]['Start_Timestamp'])
out=[]
for r in last:
    n=r['Kernel_Name']; nm='diag' if 'chol_diag16_kernel' in n else ('upd' if 'update' in n else 'strip')
    out.append((nm,(int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
print('total', (int(last[-1]['End_Timestamp'])-t0)/1e3)
upd=[d for n,s,d in out if n=='upd']; st=[d for n,s,d in out if n=='strip']
print('update launches us:', [round(x,1) for x in upd])
print('strip launches us (pairs):', [round(x,1) for x in st[:8]], '...', [round(x,1) for x in st[-8:]])
print('sum upd', round(sum(upd),1), 'sum strips', round(sum(st),1), 'diag0', [round(d,1) for n,s,d in out if n=='diag'])
PY
done
