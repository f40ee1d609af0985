set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_20
mkdir -p $O
bash tools/profile_round.sh r06_run4 > $O/profile_round.log 2>&1 || { tail -30 $O/profile_round.log; cat gpurun_out/prof_r06_run4/progress.log; exit 1; }
tail -12 $O/profile_round.log
bash tools/diag/per_step_launches.sh 40 > $O/per_step_launches.txt 2>&1 || { tail -20 $O/per_step_launches.txt; exit 1; }
cut -c1-140 $O/per_step_launches.txt | head -45
