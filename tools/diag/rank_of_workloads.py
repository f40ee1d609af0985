"""Numerical rank of S (pivoted Cholesky to 1e-16 max diag, csrc/eig.hip) at the bench workloads, with the chain's time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from vmc_pde_amd import _native as nat
dev = torch.device("cuda", 0)
for wl in sys.argv[1:] or ["C2", "C4"]:
    vs, eq, tdvp, theta = bench.build_problem(dev, wl)
    N = bench.WORKLOADS[wl]["n_samp"]
    tdvp.diagonalShift = 0.0
    tdvp(theta, 0.0, evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
    S = tdvp.S0; P = S.shape[0]; ldl = (P + 63) // 64 * 64
    LT = torch.empty((P, ldl), dtype=torch.float64, device=dev); piv = torch.zeros(P, dtype=torch.int32, device=dev)
    state = torch.zeros(2, dtype=torch.int32, device=dev)
    ws = torch.zeros(nat.lib().vmcpde_pivoted_cholesky_workspace_bytes(P) // 8, dtype=torch.float64, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    nat.call("vmcpde_pivoted_cholesky", nat.ptr(S), S.stride(0), P, 1e-16, 0, P, P, nat.ptr(LT), ldl, nat.ptr(piv), nat.ptr(state), nat.ptr(ws), ws.numel() * 8, nat.stream())
    torch.cuda.synchronize(); t1 = time.perf_counter()
    r = int(state[0])
    dk = torch.stack([LT[k, int(piv[k])] ** 2 for k in range(0, r, max(1, r // 12))])
    print(f"{wl}: P {P}, rank at 1e-16 max diag: {r} ({1e3 * (t1 - t0):.1f} ms for {P} launches); pivots d_k / d_0 at k = 0, {max(1, r // 12)}, ...: "
          + " ".join(f"{float(v / dk[0]):.1e}" for v in dk), flush=True)
