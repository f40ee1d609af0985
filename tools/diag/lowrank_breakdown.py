"""Stage times of the rank-revealing spectral decomposition (TDVP._eigh_lowrank) on the model's own S: HIP events around the pivoted
Cholesky chain, the Householder chain, the compact-WY GEMMs, the Rayleigh-Ritz core and the assembly of (ev, V)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from vmc_pde_amd import _native as nat
import vmc_pde_amd.tdvp as T
dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
vs, eq, tdvp, theta = bench.build_problem(dev, wl)
N = bench.WORKLOADS[wl]["n_samp"]
tdvp.solver, tdvp.diagonalShift = "eigh", 0.0
if len(sys.argv) > 2:
    tdvp.spectralEngine = sys.argv[2]              # "lowrank": no rank cap
tdvp(theta, 0.0, evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
S = tdvp.S0
P = S.shape[0]
print(wl, "P", P, tdvp.spectralInfo, flush=True)
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): out = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out
ms, _ = timed(lambda: tdvp._eigh_lowrank(S))
print(f"whole engine {ms:.2f} ms", flush=True)
w = tdvp._buf_lr[("lowrank", P, str(S.device))]
r = tdvp.spectralInfo["rank"]; cap = w["LT"].shape[0]; ldl = w["ldl"]
def pc():
    nat.call("vmcpde_pivoted_cholesky_blocked", nat.ptr(S), S.stride(0), P, T.LOWRANK_TOL, 1, (r + 12) // 13 + 2, cap, nat.ptr(w["LT"]), ldl, nat.ptr(w["piv"]),
             nat.ptr(w["state"]), nat.ptr(w["ws"]), w["ws"].numel() * 8, nat.stream())
ms_pc = timed(pc)[0]
print(f"blocked pivoted Cholesky, {(r + 12) // 13 + 2} block steps (2 launches each): {ms_pc:.2f} ms -> rank {int(w['state'][0])} done {int(w['state'][1])}", flush=True)
def hq():
    pc()
    nat.call("vmcpde_householder_rows", nat.ptr(w["LT"]), ldl, P, r, nat.ptr(w["YT"]), nat.ptr(w["tau"]), nat.ptr(w["hq"]), nat.stream())
print(f"  + Householder rows, {r} launches: {timed(hq)[0]:.2f} ms", flush=True)
Y = w["YT"][:r, :P]; tau = w["tau"]
def wy():
    Tinv = torch.triu(Y @ Y.T, 1); Tinv.diagonal().copy_(1.0 / tau[:r])
    X = torch.linalg.solve_triangular(Tinv, torch.eye(r, dtype=torch.float64, device=dev), upper=True) @ Y
    Q = torch.matmul(Y.T, X).neg_(); Q.diagonal().add_(1.0)
    return Q
ms, Q = timed(wy); print(f"compact WY -> Q (P x P): {ms:.2f} ms", flush=True)
Q1 = Q[:, :r]
ms, Tm = timed(lambda: Q1.T @ (S @ Q1)); print(f"Q1^T S Q1: {ms:.2f} ms", flush=True)
Tm = 0.5 * (Tm + Tm.T)
ms, (wv, W) = timed(lambda: torch.linalg.eigh(Tm)); print(f"eigh r x r (r = {r}): {ms:.2f} ms", flush=True)
def asm():
    Q[:, :r] = Q1 @ W
    vals = torch.cat([wv, torch.zeros(P - r, dtype=torch.float64, device=dev)])
    order = torch.argsort(vals, stable=True)
    return vals[order], Q.index_select(1, order)
print(f"assemble (ev, V): {timed(asm)[0]:.2f} ms", flush=True)
