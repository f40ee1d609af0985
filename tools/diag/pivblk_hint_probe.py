"""Blocked pivoted Cholesky with and without pivot hints (vmcpde_lowrank_hint) on two consecutive S of a workload: block steps used,
rank, what is left of S, and the time of the chain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from vmc_pde_amd import _native as nat
dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
vs, eq, tdvp, theta = bench.build_problem(dev, wl)
N = bench.WORKLOADS[wl]["n_samp"]
tdvp.choleskyEngine = "dense"
Ss = []
for it in range(3):
    up, _ = tdvp(theta, 0.0, evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
    Ss.append(tdvp.S0.clone())
    if it >= 1:
        theta = theta + 1e-4 * up       # (S1 -> S2: new samples only; S2 -> S3: an Euler step of 1e-4 as well)
P = Ss[0].shape[0]; ldl = (P + 63) // 64 * 64; cap = min(768, P)
LT = torch.zeros((cap, ldl), dtype=torch.float64, device=dev); piv = torch.zeros(cap, dtype=torch.int32, device=dev)
state = torch.zeros(2, dtype=torch.int32, device=dev)
ws = torch.zeros(nat.lib().vmcpde_pivoted_cholesky_blocked_workspace_bytes(P) // 8, dtype=torch.float64, device=dev)

def factor(S, nblocks=60):
    LT.zero_()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    nat.call("vmcpde_pivoted_cholesky_blocked", nat.ptr(S), S.stride(0), P, 1e-14, 1, nblocks, cap, nat.ptr(LT), ldl, nat.ptr(piv), nat.ptr(state),
             nat.ptr(ws), ws.numel() * 8, nat.stream())
    torch.cuda.synchronize(); t1 = time.perf_counter()
    r, done = int(state[0]), int(state[1])
    steps = int(ws[:4].view(torch.int32)[7])
    L = LT[:r, :P]
    R = S - L.T @ L
    return dict(rank=r, done=done, steps=steps, ms=1e3 * (t1 - t0), trace_left=float(ws[1]), resid_max=float(R.abs().max() / S.diagonal().max()),
                min_diag_R=float(R.diagonal().min() / S.diagonal().max()))

print("S1, no hint :", factor(Ss[0]))
print("S1, no hint :", factor(Ss[0]))
nat.call("vmcpde_lowrank_hint", nat.ptr(piv), nat.ptr(state), P, nat.ptr(ws), ws.numel() * 8, nat.stream())
print("S1, own hint:", factor(Ss[0]))
nat.call("vmcpde_lowrank_hint", nat.ptr(piv), nat.ptr(state), P, nat.ptr(ws), ws.numel() * 8, nat.stream())
print("S2, S1 hint :", factor(Ss[1]))
nat.call("vmcpde_lowrank_hint", None, None, P, nat.ptr(ws), ws.numel() * 8, nat.stream())
print("S2, no hint :", factor(Ss[1]))
nat.call("vmcpde_lowrank_hint", nat.ptr(piv), nat.ptr(state), P, nat.ptr(ws), ws.numel() * 8, nat.stream())
print("S3, S2 hint :", factor(Ss[2]))
nat.call("vmcpde_lowrank_hint", nat.ptr(piv), nat.ptr(state), P, nat.ptr(ws), ws.numel() * 8, nat.stream())
print("S3, S3 hint, 12 block steps launched:", factor(Ss[2], 12))
nat.call("vmcpde_lowrank_hint", None, None, P, nat.ptr(ws), ws.numel() * 8, nat.stream())
print("S3, no hint :", factor(Ss[2]))
print("S3, no hint, 19 block steps launched:", factor(Ss[2], 19))
