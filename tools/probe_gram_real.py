"""Timing probe: the Gram kernel on the REAL Jacobian of the bench problem (vs tools/probe_gram.py on N(0,1) data)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from vmc_pde_amd import _native as nat
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
vs, eq, tdvp, theta = bench.build_problem(dev, "C2")
N = 65536
x, lp = vs.sample(N)
E, J, lp2 = eq(vs, x, 0.0)
JT, ldj = vs.jt_buffer_of(J)
P = vs.numParameters
print("JT stats: absmax", float(JT.abs().max()), "frac zero", float((JT[:P, :N] == 0).double().mean()), "frac |x|<1e-300", float((JT[:P,:N].abs() < 1e-300).double().mean()),
      "row rms min/median/max", [float(v) for v in torch.quantile(JT[:P,:N].pow(2).mean(1).sqrt(), torch.tensor([0.0,0.5,1.0], dtype=torch.float64, device=dev))])
packed = torch.empty(nat.lib().vmcpde_gram_packed_doubles(P), dtype=torch.float64, device=dev)
need = nat.lib().vmcpde_gram_workspace_bytes(P, N)
ws = torch.empty(need // 8, dtype=torch.float64, device=dev)
def run(buf, label):
    ts = []
    for it in range(13):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        nat.call("vmcpde_gram", nat.ptr(buf), ldj, 0, P, N, None, None, nat.ptr(packed), nat.ptr(ws), ws.numel() * 8, nat.stream())
        e1.record(); torch.cuda.synchronize()
        if it: ts.append(e0.elapsed_time(e1))
    print(label, "median %.3f ms" % sorted(ts)[len(ts)//2], flush=True)
run(JT, "real J      ")
R = torch.randn_like(JT)
run(R, "N(0,1)      ")
run(JT * 0, "zeros       ")
scale = JT[:P,:N].pow(2).mean(1).sqrt().clamp_min(1e-300)
JTn = JT.clone(); JTn[:P,:N] /= scale[:, None]
run(JTn, "real J, rows normalised")
