# 16-wide schedule of the diagonal block (VMCPDE_CHOL_DIAG16=1, default) against the 32-wide one (=0)
for P in 640 4213 16181; do for m in 1 0; do echo "P=$P diag16=$m"; VMCPDE_CHOL_DIAG16=$m timeout -k 10 120 python tools/probe_cholesky.py $P 2>&1 | tail -1 || exit 1; done; done
timeout -k 10 300 python -m pytest tests/test_gpu_solve.py tests/test_gpu_large_configs.py -m gpu -q 2>&1 | tail -3
