# where flow_eval's time goes: timing-only builds without the activation function / without the Jacobian stores
#   python -m vmc_pde_amd.build --variant notanh -DVMC_FLOW_DIAG_NO_TANH;  ... nostore -DVMC_FLOW_DIAG_NO_STORE;
#   ... neither (both);  ... nosync -DVMC_FLOW_DIAG_NO_SYNC (no workgroup barriers);  ... bare (all three)
for v in "" notanh nostore neither nosync bare; do
  if [ -n "$v" ]; then export VMCPDE_LIB=$PWD/vmc_pde_amd/_lib/libvmcpde_$v.so; else unset VMCPDE_LIB; fi
  echo "variant '$v'"; KEEP_ACT=0 python tools/probe_flow.py C2 2>&1 | tail -1
done
