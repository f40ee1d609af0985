set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_12
mkdir -p $O
bash tools/diag/per_step_launches.sh 40 > $O/per_step_launches.txt 2>&1 || { tail -20 $O/per_step_launches.txt; exit 1; }
cat $O/per_step_launches.txt | cut -c1-140
bash tools/diag/pmc_flow_stalls.sh > $O/flow_eval_pmc_stalls.txt 2>&1 || { tail -20 $O/flow_eval_pmc_stalls.txt; exit 1; }
cat $O/flow_eval_pmc_stalls.txt
