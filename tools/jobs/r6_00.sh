set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6_00
O=gpurun_out/r6_00
python -m pytest tests/test_gpu_lowrank_eigh.py -x -q -m gpu -k "stedc or tridiag_core" > $O/t_stedc.log 2>&1 || { tail -30 $O/t_stedc.log; exit 1; }
tail -3 $O/t_stedc.log
timeout -k 10 500 python tools/diag/pipelined_chol_probe.py > $O/probe_c4.log 2>&1 || { tail -30 $O/probe_c4.log; exit 1; }
cat $O/probe_c4.log
for leg in "--engine lowrank" "--engine dense" "--solver cholesky" "--engine auto --snr"; do
  tag=$(echo $leg | tr -d ' -' )
  timeout -k 10 300 python tools/spectral_trajectory_attribution.py --leg product $leg --samples 4096 --steps 64 --out $O/attr_$tag.npz > $O/attr_$tag.log 2>&1 || { tail -30 $O/attr_$tag.log; exit 1; }
  tail -3 $O/attr_$tag.log
done
SOLVER=eigh ENGINE=dense T_END=1.0 timeout -k 10 300 python tools/run_headline_trajectory.py > $O/headline_traj_eigh_dense.txt 2>&1 || { tail -30 $O/headline_traj_eigh_dense.txt; exit 1; }
cat $O/headline_traj_eigh_dense.txt
SOLVER=eigh ENGINE=auto USE_SNR=1 T_END=1.0 timeout -k 10 300 python tools/run_headline_trajectory.py > $O/headline_traj_eigh_snr_on.txt 2>&1 || { tail -30 $O/headline_traj_eigh_snr_on.txt; exit 1; }
cat $O/headline_traj_eigh_snr_on.txt
