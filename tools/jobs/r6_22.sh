set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_22
mkdir -p $O
BENCH_ARGS="--solver eigh" bash tools/diag/per_step_launches.sh 30 > $O/per_step_launches_eigh.txt 2>&1 || { tail -20 $O/per_step_launches_eigh.txt; exit 1; }
cut -c1-150 $O/per_step_launches_eigh.txt | head -70
