echo "== mfma sweep (default)"; python tools/probe_cholesky.py 4213 2>&1 | tail -2
echo "== readlane sweep"; VMCPDE_LIB=$PWD/vmc_pde_amd/_lib/libvmcpde_readlane.so python tools/probe_cholesky.py 4213 2>&1 | tail -2
timeout -k 10 300 python -m pytest tests/test_gpu_solve.py -m gpu -q -x 2>&1 | tail -3
python tools/probe_cholesky.py 16181 2>&1 | tail -1
VMCPDE_LIB=$PWD/vmc_pde_amd/_lib/libvmcpde_readlane.so python tools/probe_cholesky.py 16181 2>&1 | tail -1
