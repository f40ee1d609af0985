# one-launch-per-panel Cholesky (VMCPDE_CHOL_PANEL=1, default) against the three-launch schedule (=0)
for P in 300 640 1000 4213; do
  echo "P=$P panel"; timeout -k 10 120 python tools/probe_cholesky.py $P 2>&1 | tail -2 || exit 1
  echo "P=$P multi-launch"; VMCPDE_CHOL_PANEL=0 timeout -k 10 120 python tools/probe_cholesky.py $P 2>&1 | tail -1 || exit 1
done
