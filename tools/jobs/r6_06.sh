set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_06
mkdir -p $O
bash tools/profile_round.sh r06_run1 > $O/profile_round.log 2>&1 || { tail -30 $O/profile_round.log; cat gpurun_out/prof_r06_run1/progress.log; exit 1; }
tail -40 $O/profile_round.log
python bench.py --steps 1000 --warmup 10 --no-cpu-baseline --no-beside > $O/bench_1000steps.json 2> $O/bench_1000steps.err || { tail -20 $O/bench_1000steps.err; exit 1; }
python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-beside > $O/bench_300steps.json 2> $O/bench_300steps.err || { tail -20 $O/bench_300steps.err; exit 1; }
python bench.py --jacobian f32 --steps 20 --warmup 5 --no-cpu-baseline --no-beside > $O/bench_f32.json 2> $O/bench_f32.err || { tail -20 $O/bench_f32.err; exit 1; }
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6_06/bench*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], round(d['value'],3), round(d['ms_per_step'],3), d['rank_stage_ms'], d['roofline']['frac'])
PY
timeout -k 10 200 python tools/soak_adaptive_headline.py > $O/soak_adaptive_heun.txt 2>&1 || { tail -20 $O/soak_adaptive_heun.txt; exit 1; }
tail -5 $O/soak_adaptive_heun.txt
timeout -k 10 200 python tools/run_headline_trajectory.py 1e-4 C4 > $O/trajectory_C4.txt 2>&1 || { tail -20 $O/trajectory_C4.txt; exit 1; }
tail -4 $O/trajectory_C4.txt
SOLVER=eigh READ_SNR=1 timeout -k 10 200 python tools/run_headline_trajectory.py > $O/headline_trajectory_eigh_snr.txt 2>&1 || { tail -20 $O/headline_trajectory_eigh_snr.txt; exit 1; }
tail -3 $O/headline_trajectory_eigh_snr.txt
