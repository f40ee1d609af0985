set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_08
mkdir -p $O
for wl in C2 C4 C3; do
  for v in "" spec; do
    if [ -n "$v" ]; then export VMCPDE_LIB=$PWD/vmc_pde_amd/_lib/libvmcpde_$v.so; else unset VMCPDE_LIB; fi
    echo "== $wl variant '$v'" >> $O/flow_spec_ab.txt; KEEP_ACT=0 python tools/probe_flow.py $wl 2>&1 | tail -2 >> $O/flow_spec_ab.txt
  done
done
unset VMCPDE_LIB
cat $O/flow_spec_ab.txt
