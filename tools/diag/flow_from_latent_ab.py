"""Round 6: time of the one-launch sampling + evaluation (EvolutionEquation.from_latent) under the flow kernels' plan knobs (VMCPDE_FLOW_CAP, _JB, _LANES,
_FAST are read per call).  Usage: python tools/diag/flow_from_latent_ab.py [C2|C4|C3]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
dev = torch.device("cuda", 0)
vs, eq, tdvp, theta = bench.build_problem(dev, wl)
N = bench.WORKLOADS[wl]["n_samp"]


def timed(reps=7):
    ts = []
    for _ in range(reps + 1):
        z = vs.draw_latent(N)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eq.from_latent(vs, z, 0.0)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = sorted(ts[1:])
    return ts[len(ts) // 2], ts[0]


for name, env in (("default", {}), ("CAP=0", {"VMCPDE_FLOW_CAP": "0"}), ("JB=5", {"VMCPDE_FLOW_JB": "5"}), ("JB=5 CAP=0", {"VMCPDE_FLOW_JB": "5", "VMCPDE_FLOW_CAP": "0"}),
                  ("LANES=32", {"VMCPDE_FLOW_LANES": "32"}), ("FAST=0", {"VMCPDE_FLOW_FAST": "0"})):
    for k in ("VMCPDE_FLOW_CAP", "VMCPDE_FLOW_JB", "VMCPDE_FLOW_LANES", "VMCPDE_FLOW_FAST"):
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        t = timed()
        print(f"{wl} {name:12s} from_latent (flow launch + rhs) {t[0]:.3f} / {t[1]:.3f} ms", flush=True)
    except Exception as exc:
        print(f"{wl} {name:12s} failed: {str(exc)[:100]}", flush=True)
