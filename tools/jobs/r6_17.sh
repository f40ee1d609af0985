set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_17
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
D=$GRAFT_REPO_ROOT/gpurun_out/timeline_r6
rm -rf $D
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-beside > $D.log 2>&1 || { tail -5 $D.log; exit 1; }
cd $GRAFT_REPO_ROOT
f=$(find $D -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py $f --step 9 --list > $O/step_timeline.txt 2>&1 || { tail $O/step_timeline.txt; exit 1; }
cut -c1-150 $O/step_timeline.txt
