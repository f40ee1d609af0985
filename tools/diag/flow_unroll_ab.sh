# dense-layer loop unroll factor of the flow kernels (builds: python -m vmc_pde_amd.build --variant unrK -DVMC_FLOW_UNROLL=K)
R=${GRAFT_REPO_ROOT:-$PWD}
for wl in ${@:-C2 C3 C4}; do
  for v in prevflow unr1 unr2 unr3; do
    KEEP_ACT=0 VMCPDE_LIB=$R/vmc_pde_amd/_lib/libvmcpde_$v.so python3 $R/tools/probe_flow.py $wl 2>/dev/null | tail -1 | sed "s/^/$v /"
  done
done
