# upper bounds for the Gram main loop (timing-only builds, results wrong): without the K-step barrier, and without barrier + DMA wait
#   python -m vmc_pde_amd.build --variant gnobar -DVMC_GRAM_DIAG_NO_BARRIER;  ... gnowait -DVMC_GRAM_DIAG_NO_BARRIER -DVMC_GRAM_DIAG_NO_WAIT
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
  for v in "" gnobar gnowait; do
    if [ -n "$v" ]; then export VMCPDE_LIB=$R/vmc_pde_amd/_lib/libvmcpde_$v.so; else unset VMCPDE_LIB; fi
    python3 $R/tools/probe_gram.py 4213 65536 7 nocentre 2>/dev/null | sed "s/^/[$v] /" | cut -c1-120
  done
done
