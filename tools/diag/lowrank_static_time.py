"""Would a fixed-shape, host-read-free low-rank Cholesky solve beat the dense factorisation at C2?  GPU time of the pipeline with
static shapes (rank padded to rmax, zero rows) and the library's own small Cholesky (csrc/solve.hip) instead of rocSOLVER's."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from vmc_pde_amd import _native as nat
import vmc_pde_amd.tdvp as T
dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
vs, eq, tdvp, theta = bench.build_problem(dev, wl)
N = bench.WORKLOADS[wl]["n_samp"]
kw = dict(evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
tdvp.choleskyEngine = "dense"
for _ in range(2):
    u_d, _ = tdvp(theta, 0.0, **kw)
b = tdvp._lazy["b"]; P = tdvp.F0.numel(); F = b["F"][:P].clone(); S0 = b["S"][:P, :P]
D = b["diagS"] * float(tdvp.diagonalShift) + tdvp._floor
rmax = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 22
w = tdvp._lowrank_buffers(P, dev, max(rmax, P // 6))
G = torch.zeros((rmax, rmax), dtype=torch.float64, device=dev)
info = torch.zeros(1, dtype=torch.int32, device=dev)
need = nat.lib().vmcpde_chol_workspace_bytes(rmax)
cws = torch.empty(need // 8, dtype=torch.float64, device=dev)
Su = torch.empty(P, dtype=torch.float64, device=dev)
t2 = torch.empty(rmax, dtype=torch.float64, device=dev)
def pipeline(refine=2):
    w["LT"][:rmax].zero_()
    nat.call("vmcpde_pivoted_cholesky_blocked", nat.ptr(S0), S0.stride(0), P, T.LOWRANK_TOL, 1, nsteps, rmax, nat.ptr(w["LT"]), w["ldl"], nat.ptr(w["piv"]),
             nat.ptr(w["state"]), nat.ptr(w["ws"]), w["ws"].numel() * 8, nat.stream())
    LTr = w["LT"][:rmax, :P]
    Ls = LTr / D
    torch.mm(Ls, LTr.T, out=G); G.diagonal().add_(1.0)
    nat.call("vmcpde_cholesky", nat.ptr(G), rmax, rmax, nat.ptr(info), nat.ptr(cws), need, nat.stream())
    def minv(v):
        t1 = Ls @ v
        nat.call("vmcpde_chol_solve", nat.ptr(G), rmax, rmax, nat.ptr(t1), nat.ptr(t2), nat.ptr(cws), need, nat.stream())
        return torch.addmv(v / D, Ls.T, t2, alpha=-1.0)
    u = minv(F)
    for _ in range(refine):
        nat.call("vmcpde_gemv", nat.ptr(b["S"]), b["Ppad"], P, P, 0, nat.ptr(u), nat.ptr(Su), nat.stream())
        u = u + minv(F - Su - D * u)
    return u
def gpu_ms(f, reps=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): out = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out
for refine in (1, 2):
    ms, u = gpu_ms(lambda: pipeline(refine))
    nF = float(torch.linalg.norm(F))
    print(f"rmax {rmax}, {nsteps} block steps, {refine} refinement(s): {ms:.3f} ms per solve (GPU, back to back); state {w['state'].tolist()}; residual "
          f"{float(torch.linalg.norm(S0 @ u + D * u - F)) / nF:.2e}; |u - u_dense|_A {float(torch.linalg.norm(S0 @ (u - u_d) + D * (u - u_d))) / nF:.2e}", flush=True)
A = b["A"]; Ppad = b["Ppad"]
def dense():
    tdvp._lazy  # the dense factorisation on a fresh copy of A each time
    A2.copy_(A0)
    nat.call("vmcpde_cholesky", nat.ptr(A2), Ppad, P + 1, nat.ptr(info), nat.ptr(b["chol_ws"]), b["chol_ws"].numel() * 8, nat.stream())
A0 = torch.zeros_like(A); A0[:P, :P] = S0 + torch.diag(D); A0[P, :P] = F; A0[P, P] = 1e200
A2 = A0.clone()
ms_copy, _ = gpu_ms(lambda: A2.copy_(A0))
ms, _ = gpu_ms(dense)
print(f"dense bordered factorisation: {ms - ms_copy:.3f} ms (+ 0.13 ms back substitution)")
