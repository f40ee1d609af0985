"""Time of a fixed number of block steps of the blocked pivoted Cholesky on the model's S, with the kernel's timing diagnostics
(VMCPDE_PIVBLK_DIAG: 1 one elimination per block only, 2 no row substitution -- wrong results)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from vmc_pde_amd import _native as nat
import vmc_pde_amd.tdvp as T
dev = torch.device("cuda", 0)
vs, eq, tdvp, theta = bench.build_problem(dev, "C2")
tdvp.choleskyEngine = "dense"
tdvp(theta, 0.0, evolutionEq=eq, psi=vs, nSamplesTDVP=65536, nSamplesObs=65536, timings=None, observables=False)
b = tdvp._lazy["b"]; P = tdvp.F0.numel(); S0 = b["S"][:P, :P]
w = tdvp._lowrank_buffers(P, dev, P // 6)
def run(nb):
    nat.call("vmcpde_pivoted_cholesky_blocked", nat.ptr(S0), S0.stride(0), P, T.LOWRANK_TOL, 1, nb, 256, nat.ptr(w["LT"]), w["ldl"], nat.ptr(w["piv"]),
             nat.ptr(w["state"]), nat.ptr(w["ws"]), w["ws"].numel() * 8, nat.stream())
for nb in (8, 16):
    run(nb); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run(nb)
    e1.record(); torch.cuda.synchronize()
    print(f"diag {os.environ.get('VMCPDE_PIVBLK_DIAG', '0')}: {nb} block steps {e0.elapsed_time(e1) / 20 * 1e3:.0f} us -> rank {int(w['state'][0])}", flush=True)
