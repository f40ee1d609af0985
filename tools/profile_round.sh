#!/bin/bash
# Profiles of the default bench (C2) for profiles/: per-kernel times (rocprofv3 --kernel-trace --stats) and, in separate
# passes as MI355X_MICROARCH.md prescribes, the PMC counters of the Gram kernel (HBM traffic, MFMA-busy cycles).
#   usage (on the GPU box):  bash tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/{stats,pmc_*}/ + summary json
set -u
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "[$(date +%T)] kernel stats" >> $OUT/progress.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-beside > $OUT/stats.log 2>&1 || { echo "stats pass failed" >> $OUT/progress.log; exit 1; }
for c in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
  n=$(echo $c | tr ' ' '+')
  echo "[$(date +%T)] pmc $n" >> $OUT/progress.log
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$n -- python3 $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-beside > $OUT/pmc_$n.log 2>&1 || { echo "pmc $n failed" >> $OUT/progress.log; exit 1; }
done
echo "[$(date +%T)] summarise" >> $OUT/progress.log
python3 $R/tools/summarise_profile.py $OUT $TAG
