# A/B on one box: current flow kernels against the previous build (python -m vmc_pde_amd.build --variant prevflow, made before the change)
R=${GRAFT_REPO_ROOT:-$PWD}
for wl in ${@:-C2 C3 C4}; do
  for rep in 1 2; do
    KEEP_ACT=0 python3 $R/tools/probe_flow.py $wl 2>/dev/null | tail -1 | sed 's/^/new  /'
    KEEP_ACT=0 VMCPDE_LIB=$R/vmc_pde_amd/_lib/libvmcpde_prevflow.so python3 $R/tools/probe_flow.py $wl 2>/dev/null | tail -1 | sed 's/^/prev /'
  done
done
