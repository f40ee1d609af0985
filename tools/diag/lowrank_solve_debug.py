"""Contraction of the iterative refinement of the low-rank Cholesky solve (TDVP._solve_lowrank) on the model's own (S, F)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import vmc_pde_amd.tdvp as T
dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
vs, eq, tdvp, theta = bench.build_problem(dev, wl)
N = bench.WORKLOADS[wl]["n_samp"]
tdvp.choleskyEngine = "dense"
kw = dict(evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
for it in range(3):
    u_d, _ = tdvp(theta, 0.0, **kw)
    b = tdvp._lazy["b"]; P = u_d.numel()
    S0 = b["S"][:P, :P].clone(); F = b["F"][:P].clone()
    D = b["diagS"] * float(tdvp.diagonalShift) + tdvp._floor
    cap = P // 6
    w = tdvp._lowrank_buffers(P, dev, cap)
    r, done, d0, left, dmin = tdvp._pivoted_factor(S0, w, cap, extra=D.min())
    LTr = w["LT"][:r, :P]
    Rm = S0 - LTr.T @ LTr
    print(f"call {it}: rank {r} done {done} trace_left/dmin {left/dmin:.2e}; |R|_2 / dmin {float(torch.linalg.matrix_norm(Rm, 2))/dmin:.2e} |R|_F/dmin {float(torch.linalg.norm(Rm))/dmin:.2e} min diag R / d0 {float(torch.diagonal(Rm).min())/d0:.2e} max {float(torch.diagonal(Rm).max())/d0:.2e}", flush=True)
    Ls = LTr / D
    G = Ls @ LTr.T; G.diagonal().add_(1.0)
    Gc = torch.linalg.cholesky(G)
    print(f"   cond(G) {float(torch.linalg.cond(G)):.2e}")
    def minv(v):
        t = torch.cholesky_solve((Ls @ v)[:, None], Gc)[:, 0]
        return v / D - Ls.T @ t
    u = minv(F); nF = float(torch.linalg.norm(F))
    for k in range(5):
        res = F - S0 @ u - D * u
        print(f"   refinement {k}: residual {float(torch.linalg.norm(res))/nF:.2e}   |u - u_dense|_A {float(torch.linalg.norm(S0 @ (u-u_d) + D*(u-u_d)))/nF:.2e}")
        u = u + minv(res)
    print(f"   dense residual {float(torch.linalg.norm(F - S0 @ u_d - D * u_d))/nF:.2e}")
    # where does L L^T overshoot S?  the worst row, column by column
    i = int(torch.argmin(torch.diagonal(Rm)))
    piv = w["piv"][:r].cpu().numpy().tolist()
    acc = float(S0[i, i]); Lrow = LTr[:, i].cpu().numpy()
    print(f"   worst row {i}: S_ii/d0 {float(S0[i,i])/d0:.3e}, is pivot at column {piv.index(i) if i in piv else None}")
    run = float(S0[i, i])
    for k in range(r):
        before = run; run -= Lrow[k] ** 2
        if run < -1e-12 * d0 and before >= -1e-12 * d0 or abs(Lrow[k] ** 2) > 0.2 * max(before, 1e-300) and before < 1e-3 * float(S0[i, i]):
            print(f"      col {k} (pivot row {piv[k]}, l_kk^2/d0 {float(LTr[k, piv[k]])**2/d0:.2e}): remaining before {before/d0:.3e} L_ik^2/d0 {Lrow[k]**2/d0:.3e} after {run/d0:.3e}")
    break
