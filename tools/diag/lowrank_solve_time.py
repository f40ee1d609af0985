"""Where the time of the low-rank Cholesky solve goes (TDVP._solve_lowrank) at a bench workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
vs, eq, tdvp, theta = bench.build_problem(dev, wl)
N = bench.WORKLOADS[wl]["n_samp"]
kw = dict(evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
for _ in range(3):
    tdvp(theta, 0.0, **kw)
print(tdvp.choleskyInfo)
b = tdvp._lazy["b"]; P = tdvp.F0.numel(); F = b["F"][:P]
S0 = b["S"][:P, :P]
D = b["diagS"] * float(tdvp.diagonalShift) + tdvp._floor
cap = P // 6
w = tdvp._lowrank_buffers(P, dev, cap)
def timed(f, reps=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): out = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3, out
calls = [0]
orig = tdvp._pivoted_factor
ms, out = timed(lambda: orig(S0, w, cap, extra=D.min()))
print(f"pivoted factor (blocked, incl. host reads): {ms:.3f} ms -> rank {out[0]}, done {out[1]}")
from vmc_pde_amd import _native as nat
import vmc_pde_amd.tdvp as T
def raw(nb):
    nat.call("vmcpde_pivoted_cholesky_blocked", nat.ptr(S0), S0.stride(0), P, T.LOWRANK_TOL, 1, nb, cap, nat.ptr(w["LT"]), w["ldl"], nat.ptr(w["piv"]),
             nat.ptr(w["state"]), nat.ptr(w["ws"]), w["ws"].numel() * 8, nat.stream())
for nb in (4, 8, 12, 16):
    ms, _ = timed(lambda: raw(nb)); torch.cuda.synchronize()
    print(f"   {nb} block steps, no host read: {ms:.3f} ms -> rank {int(w['state'][0])} done {int(w['state'][1])}")
upd = torch.empty(P, dtype=torch.float64, device=dev)
ms, _ = timed(lambda: tdvp._solve_lowrank(b, F, P, upd))
print(f"whole _solve_lowrank: {ms:.3f} ms")
r = out[0]
LTr = w["LT"][:r, :P]
def wood():
    Ls = LTr / D; G = Ls @ LTr.T; G.diagonal().add_(1.0); Gc = torch.linalg.cholesky(G)
    return Ls, Gc
ms, (Ls, Gc) = timed(wood); print(f"Ls, G, cholesky(G): {ms:.3f} ms")
def minv(v):
    t = torch.cholesky_solve((Ls @ v)[:, None], Gc)[:, 0]
    return v / D - Ls.T @ t
ms, _ = timed(lambda: minv(F)); print(f"one application of M^-1: {ms:.3f} ms")
ms, _ = timed(lambda: torch.linalg.cholesky(Gc @ Gc.T)); print(f"(cholesky r x r alone incl. a GEMM: {ms:.3f} ms)")
tdvp.choleskyEngine = "dense"
ms, _ = timed(lambda: tdvp(theta, 0.0, **kw), 5); print(f"TDVP call, dense engine: {ms:.3f} ms")
tdvp.choleskyEngine = "auto"
ms, _ = timed(lambda: tdvp(theta, 0.0, **kw), 5); print(f"TDVP call, auto engine: {ms:.3f} ms")
