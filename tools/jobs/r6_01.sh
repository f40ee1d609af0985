set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_01
mkdir -p $O
A="python tools/spectral_trajectory_attribution.py --leg product --samples 4096"
for tol in 1e-6 1e-3; do
  for leg in "--engine lowrank" "--engine dense" "--solver cholesky" "--engine auto --snr"; do
    tag=$(echo $leg | tr -d ' -' )_svd$tol
    timeout -k 10 300 $A $leg --svd-tol $tol --steps 64 --out $O/attr_$tag.npz > $O/attr_$tag.log 2>&1 || { tail -30 $O/attr_$tag.log; exit 1; }
    grep -v Warning $O/attr_$tag.log | tail -4
  done
done
SOLVER=eigh ENGINE=dense T_END=1.0 timeout -k 10 300 python tools/run_headline_trajectory.py > $O/headline_traj_eigh_dense.txt 2>&1 || { tail -30 $O/headline_traj_eigh_dense.txt; exit 1; }
cat $O/headline_traj_eigh_dense.txt
SOLVER=eigh ENGINE=auto T_END=1.0 timeout -k 10 300 python tools/run_headline_trajectory.py > $O/headline_traj_eigh_auto.txt 2>&1 || { tail -30 $O/headline_traj_eigh_auto.txt; exit 1; }
cat $O/headline_traj_eigh_auto.txt
SOLVER=eigh ENGINE=auto USE_SNR=1 T_END=1.0 timeout -k 10 300 python tools/run_headline_trajectory.py > $O/headline_traj_eigh_snr_on.txt 2>&1 || { tail -30 $O/headline_traj_eigh_snr_on.txt; exit 1; }
cat $O/headline_traj_eigh_snr_on.txt
SOLVER=eigh ENGINE=auto SVD_TOL=1e-4 T_END=1.0 timeout -k 10 300 python tools/run_headline_trajectory.py > $O/headline_traj_eigh_svd1e-4.txt 2>&1 || { tail -30 $O/headline_traj_eigh_svd1e-4.txt; exit 1; }
cat $O/headline_traj_eigh_svd1e-4.txt
