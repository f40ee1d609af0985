"""Accuracy of rocSOLVER's syevd against host LAPACK on the actual (singular) S of the C2 problem: residual, orthogonality,
and the pseudo-inverse update each gives for the same S, F."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, bench
dev = torch.device("cuda", 0)
vs, eq, tdvp, theta = bench.build_problem(dev, "C2")
kw = dict(evolutionEq=eq, psi=vs, nSamplesTDVP=65536, nSamplesObs=65536, timings=None, observables=False)
tdvp.solver = "eigh"
tdvp(theta, 0.0, **kw)
S, F = tdvp.S.clone(), tdvp.F0.clone()
P = F.numel()
ev_d, V_d = torch.linalg.eigh(S)
ev_h, V_h = np.linalg.eigh(S.cpu().numpy())
ev_h, V_h = torch.from_numpy(ev_h).to(dev), torch.from_numpy(V_h).to(dev)
I = torch.eye(P, dtype=torch.float64, device=dev)
for name, ev, V in (("device", ev_d, V_d), ("host", ev_h, V_h)):
    print(f"{name}: |SV-VL|/|S| {float(torch.linalg.norm(S @ V - V * ev) / torch.linalg.norm(S)):.2e}  |VtV-I| {float((V.T @ V - I).abs().max()):.2e}  "
          f"lmax {float(ev[-1]):.6e}  lmin {float(ev[0]):.3e}  #>1e-6lmax {int((ev > 1e-6 * ev[-1]).sum())}")
print("max |ev_d - ev_h| / lmax", float((ev_d - ev_h).abs().max() / ev_h[-1]))
def pinv_update(ev, V, svdTol):
    rel = (ev / ev[-1]).abs()
    coef = torch.where(rel > 1e-14, 1.0 / ev, torch.zeros_like(ev)) / (1.0 + (svdTol / rel) ** 6) * (V.T @ F)
    return V @ coef
for svdTol in (1e-6, 1e-11):
    u_d, u_h = pinv_update(ev_d, V_d, svdTol), pinv_update(ev_h, V_h, svdTol)
    nF = torch.linalg.norm(F)
    print(f"svdTol {svdTol:g}: |S(u_d-u_h)|/|F| {float(torch.linalg.norm(S @ (u_d - u_h)) / nF):.2e}  |u_d-u_h|/|u_h| {float(torch.linalg.norm(u_d - u_h) / torch.linalg.norm(u_h)):.2e} "
          f" residual d {float(torch.linalg.norm(S @ u_d - F) / nF):.3e} h {float(torch.linalg.norm(S @ u_h - F) / nF):.3e}")
# and the run-to-run spread of tdvp_error with fresh samples (Monte-Carlo noise of the estimate itself)
errs = {}
for where in (None, False, None, False):
    tdvp.diagonalizeOnDevice = where
    tdvp(theta, 0.0, **kw)
    errs.setdefault(str(where), []).append(round(float(tdvp.tdvp_error), 4))
print("tdvp_error over fresh sample sets:", errs)
# same samples for every call (the sampler's stream id is its call count): any difference is in the solve path
print("-- identical samples")
res = {}
for where in (None, False, None, False):
    tdvp.diagonalizeOnDevice = where
    vs.sampler._calls = 1000
    u, _ = tdvp(theta, 0.0, **kw)
    torch.cuda.synchronize()
    res.setdefault(str(where), []).append((u.clone(), float(tdvp.tdvp_error), float(tdvp.solverResidual), tdvp.ev.clone(), tdvp.VtF.clone()))
for k, v in res.items():
    print(k, [(round(e, 6), f"{r:.3e}") for _, e, r, _, _ in v])
ud, uh = res["None"][0][0], res["False"][0][0]
print("|u_dev - u_host|/|u_host|", float(torch.linalg.norm(ud - uh) / torch.linalg.norm(uh)), " repeat dev-dev", float(torch.linalg.norm(res["None"][1][0] - ud) / torch.linalg.norm(ud)),
      " ev diff", float((res["None"][0][3] - res["False"][0][3]).abs().max()), " |VtF| diff", float((res["None"][0][4].abs() - res["False"][0][4].abs()).abs().max()))
