"""Timing probe: flow_eval / flow_sample of the C2 (or given) bench network at N samples."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
dev = torch.device("cuda", 0)
vs, eq, tdvp, theta = bench.build_problem(dev, wl)
vs.keep_activations = os.environ.get("KEEP_ACT", "1") != "0"      # 0: the reverse sweep re-evaluates the networks
N = bench.WORKLOADS[wl]["n_samp"]
x, _ = vs.sample(N)
dirs = eq.dirs if hasattr(eq, "dirs") else None
for it in range(4):
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    x, _ = vs.sample(N)
    e[1].record()
    E, J, lp = eq(vs, x, 0.0)
    e[2].record()
    torch.cuda.synchronize()
    print(f"{wl} keep_activations={vs.keep_activations}: sample {e[0].elapsed_time(e[1]):.3f} ms, Eloc+J {e[1].elapsed_time(e[2]):.3f} ms", flush=True)
