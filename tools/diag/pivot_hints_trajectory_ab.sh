set -e
for mf in 0 1; do for h in 1 0; do
  VMCPDE_PIVOT_HINTS=$h MATRIX_FREE=$mf python tools/run_headline_trajectory.py > gpurun_out/traj_mf${mf}_h$h.txt 2>&1
  echo "== matrix-free $mf hints $h"; tail -3 gpurun_out/traj_mf${mf}_h$h.txt
done; done
