# bench lines for profiles/: the driver's launch form with one rank (torchrun), then the C3 and C4 workloads
set -x
cd $GRAFT_REPO_ROOT
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_torchrun1.log 2>&1 && tail -1 gpurun_out/bench_torchrun1.log | cut -c1-300
python bench.py --workload C3 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_C3.log 2>&1 && tail -1 gpurun_out/bench_C3.log > gpurun_out/r02_run3_bench_C3.json && cut -c1-200 gpurun_out/r02_run3_bench_C3.json
python bench.py --workload C4 --no-cpu-baseline > gpurun_out/bench_C4.log 2>&1 && tail -1 gpurun_out/bench_C4.log > gpurun_out/r02_run3_bench_C4.json && cut -c1-200 gpurun_out/r02_run3_bench_C4.json
