"""Prototype (torch ops only) of the rank-revealing spectral decomposition on the model's own S: pivoted Cholesky -> Householder QR of
the factor -> Rayleigh-Ritz on its range -> full orthogonal V = [null space | Ritz vectors]; against torch.linalg.eigh (rocSOLVER)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device("cuda", 0)
def sync():
    torch.cuda.synchronize(); return time.perf_counter()
for wl in sys.argv[1:] or ["C2"]:
    vs, eq, tdvp, theta = bench.build_problem(dev, wl)
    N = bench.WORKLOADS[wl]["n_samp"]
    tdvp(theta, 0.0, evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
    S, F = tdvp.S0.contiguous().clone(), tdvp.F0.clone()
    P = S.shape[0]
    for n in (128, 256, 384, 512):
        A = torch.randn(n, n, dtype=torch.float64, device=dev); A = A @ A.T
        torch.linalg.eigh(A); t0 = sync(); torch.linalg.eigh(A); t1 = sync()
        Ac = A.cpu(); t2 = time.perf_counter(); torch.linalg.eigh(Ac); t3 = time.perf_counter()
        print(f"eigh n={n}: device {1e3*(t1-t0):.2f} ms, host {1e3*(t3-t2):.2f} ms", flush=True)
    if P < 9000:
        torch.linalg.eigh(S); t0 = sync(); ev_ref, V_ref = torch.linalg.eigh(S); t1 = sync()
        print(f"{wl}: rocSOLVER eigh P={P}: {1e3*(t1-t0):.1f} ms", flush=True)
    else:
        ev_ref = V_ref = None
    t0 = sync()
    d = torch.diagonal(S).clone(); d0 = float(d.max()); cap = 1024
    LT = torch.zeros(cap, P, dtype=torch.float64, device=dev); piv = []
    for k in range(cap):
        p = int(torch.argmax(d)); dp = float(d[p])
        if not dp > 1e-16 * d0: break
        col = S[p] - LT[:k, p] @ LT[:k]
        col /= dp ** 0.5
        LT[k] = col; d -= col * col; piv.append(p); d[piv] = 0.0
    r = len(piv); t1 = sync()
    L = LT[:r].T.contiguous()
    Qf, _ = torch.linalg.qr(L, mode="complete"); t2 = sync()
    Q1, Q2 = Qf[:, :r], Qf[:, r:]
    B = S @ Q1; T = Q1.T @ B; T = 0.5 * (T + T.T); t3 = sync()
    w, W = torch.linalg.eigh(T); t4 = sync()
    U = Q1 @ W
    ev = torch.cat([torch.zeros(P - r, dtype=torch.float64, device=dev), w])
    V = torch.cat([Q2, U], dim=1)
    order = torch.argsort(ev, stable=True); ev = ev[order]; V = V[:, order].contiguous(); t5 = sync()
    print(f"{wl}: rank {r}; pivoted Cholesky (torch loop) {1e3*(t1-t0):.1f} ms, qr(complete) {1e3*(t2-t1):.1f}, S Q1 + T {1e3*(t3-t2):.2f}, eigh(T) {1e3*(t4-t3):.2f}, assemble V {1e3*(t5-t4):.2f}", flush=True)
    lmax = float(ev[-1])
    def upd(ev, V, svdTol):
        VtF = V.T @ F; rel = (ev / ev[-1]).abs()
        inv = torch.where(rel > 1e-14, 1.0 / torch.where(ev == 0, torch.ones_like(ev), ev), torch.zeros_like(ev))
        reg = 1.0 / (1.0 + (svdTol / rel) ** 6)
        return V @ (inv * reg * VtF)
    print(f"   orth {float((V.T @ V - torch.eye(P, device=dev, dtype=torch.float64)).abs().max()):.1e}, residual |S V - V ev| / lmax {float(torch.linalg.norm(S @ V - V * ev)) / lmax:.2e}", flush=True)
    if ev_ref is not None:
        print(f"   |ev - ref|max / lmax {float((ev - ev_ref).abs().max()) / lmax:.2e}; eigh's own residual {float(torch.linalg.norm(S @ V_ref - V_ref * ev_ref)) / lmax:.2e}")
        for tol in (1e-6, 1e-11):
            u, ur = upd(ev, V, tol), upd(ev_ref, V_ref, tol)
            print(f"   svdTol {tol:g}: update rel diff {float(torch.linalg.norm(u - ur) / torch.linalg.norm(ur)):.2e}, S-norm {float(torch.linalg.norm(S @ (u - ur)) / torch.linalg.norm(F)):.2e}", flush=True)
