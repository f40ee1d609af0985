"""Diagnostic (round 3): basis-independent versions of the snr / rhoVar and stepper comparisons.
(a) cluster sums: eigenvalues chained by gaps < 1e-6 lambda_max form a cluster; sum_k rhoVar_k and sum_k VtF_k^2 over a cluster are
    invariants of its eigenspace.  Prints the largest relative deviation product vs oracle per case.
(b) Heun / AdaptiveHeun increments: (theta1 - theta0) product vs oracle in the S-norm of the oracle's first stage and projected on
    lambda > 1e-9 lambda_max, relative to |F| dt."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import vmc_oracle as O
from helpers import make_varstate, FixedSampler
import test_gpu_tdvp as T
import vmc_pde_amd as V

torch.set_num_threads(8)


def clusters(ev, tol):
    """index lists of eigenvalues (ascending) chained by gaps < tol"""
    out, cur = [], [0]
    for k in range(1, len(ev)):
        if ev[k] - ev[k - 1] < tol:
            cur.append(k)
        else:
            out.append(cur); cur = [k]
    out.append(cur)
    return out


for case in T.CASES:
    dim, depth, hidden, latent, coupling, eq, N = case
    spec, th, nrm, chi2, eo, ep = T._problem(dim, depth, hidden, latent, coupling, eq, N, lam=0.1)
    z = O.latent_from_normals(spec, th, nrm, chi2)
    to = O.TDVPOracle(useSNR=True)
    upd_o, aux = to.rhs(th, 0.2, O.VarStateOracle(spec, th), eo, z)
    vs = make_varstate(spec, th, sampler=FixedSampler(z.cuda()))
    tp = V.TDVP(useSNR=True)
    upd_p, info = tp(th.cuda(), 0.2, vs, ep, nSamplesTDVP=N, nSamplesObs=N, timings=None)
    ev = to.ev.numpy()
    for tol in (1e-6, 1e-5, 1e-4):
        cl = clusters(ev, tol * ev[-1])
        rho_p, rho_o = tp.rhoVar.cpu().numpy(), to.rhoVar.numpy()
        f_p, f_o = tp.VtF.cpu().numpy() ** 2, to.VtF.numpy() ** 2
        worst_r = worst_f = worst_s = 0.0
        sizes = []
        for c in cl:
            if ev[c[-1]] < 1e-9 * ev[-1]:
                continue            # the whole cluster is numerical null space
            sizes.append(len(c))
            r_p, r_o, s_p, s_o = rho_p[c].sum(), rho_o[c].sum(), f_p[c].sum(), f_o[c].sum()
            worst_r = max(worst_r, abs(r_p - r_o) / abs(r_o))
            worst_f = max(worst_f, abs(s_p - s_o) / max(abs(s_o), 1e-300))
            worst_s = max(worst_s, abs(np.sqrt(N * s_p / r_p) - np.sqrt(N * s_o / r_o)) / np.sqrt(N * s_o / r_o))
        print(f"{case} tol {tol:g}: {len(sizes)} clusters (max size {max(sizes)}, singletons {sizes.count(1)}) "
              f"rel dev rhoVar {worst_r:.2e} VtF^2 {worst_f:.2e} cluster-snr {worst_s:.2e}", flush=True)
    # total invariants
    print(f"    trace: sum rhoVar {abs(rho_p.sum() - rho_o.sum()) / rho_o.sum():.2e}  sum VtF^2 {abs(f_p.sum() - f_o.sum()) / f_o.sum():.2e}")
