# kernel timeline of one C2 step inside bench.py's timed loop (rocprofv3 kernel trace -> tools/step_timeline.py)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/timeline
rm -rf $D
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $D.log 2>&1
f=$(find $D -name "*kernel_trace.csv" | head -1)
python3 $R/tools/step_timeline.py $f
