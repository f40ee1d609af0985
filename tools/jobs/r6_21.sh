set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_21
mkdir -p $O
VMCPDE_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_n2_torchrun.json 2> $O/bench_n2_torchrun.err || { tail -30 $O/bench_n2_torchrun.err; exit 1; }
tail -1 $O/bench_n2_torchrun.json | cut -c1-900
VMCPDE_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 10 --warmup 3 --workload C4 > $O/bench_n2_C4.json 2> $O/bench_n2_C4.err || { tail -30 $O/bench_n2_C4.err; exit 1; }
tail -1 $O/bench_n2_C4.json | cut -c1-900
