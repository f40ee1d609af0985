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


# ---- (b) stepper increments in the S-norm of the first stage --------------------------------------------------------------
def stage_metrics(tag, y_p, y_o, th0, S0, F0, ev, Vv, dt):
    d = (y_p - th0) - (y_o - th0)
    step = y_o - th0
    nF = float(torch.linalg.norm(F0))
    W = Vv[:, ev > 1e-9 * ev[-1]]
    dW = W @ (W.T @ d)
    print(f"{tag}: dt {dt:.3e} |step| {float(torch.linalg.norm(step)):.3e} whole-vector rel {float(torch.linalg.norm(d) / torch.linalg.norm(step)):.2e} | "
          f"|S0 d|/(|F| dt) {float(torch.linalg.norm(S0 @ d)) / (nF * dt):.2e} | projected |S0 P d|/(|F| dt) {float(torch.linalg.norm(S0 @ dW)) / (nF * dt):.2e} "
          f"| |P d|/|P step| {float(torch.linalg.norm(dW) / torch.linalg.norm(W @ (W.T @ step))):.2e}", flush=True)


from vmc_pde_amd import _native as nat

dim, depth, hidden, latent, coupling, eq, N = 2, 4, (1,), "Gauss", "no_add", "diffusion", 1024
spec, th, nrm, chi2, eo, ep = T._problem(dim, depth, hidden, latent, coupling, eq, N)
g = torch.Generator().manual_seed(99)
normals = [nrm, torch.randn(N, dim, generator=g, dtype=torch.float64)]
vo, to = O.VarStateOracle(spec, th), O.TDVPOracle()
rec = {}


def f_o(y, t, k):
    z = O.latent_from_normals(spec, y, normals[k], chi2)
    u, aux = to.rhs(y, t, vo, eo, z)
    if k == 0:
        rec.update(S0=to.S0.clone(), F0=to.F0.clone(), ev=to.ev.clone(), V=to.V.clone())
    return u, aux


dt = O.fixed_stepper_dt(1e-3, 1.3, 1e-2)
y_o, _ = O.heun_step(f_o, th.clone(), 0.0, dt)


class StageSampler:
    key = 1

    def __init__(self, normals):
        self.k, self.normals = 0, normals

    def __call__(self, numSamples, dist_params, multipleOf=1):
        z = torch.empty((N, dim), dtype=torch.float64, device="cuda")
        nat.call("vmcpde_latent_from_normals", dist_params["_model"].handle, nat.ptr(dist_params["_theta"]),
                 nat.ptr(self.normals[self.k].cuda()), None, N, nat.ptr(z), nat.stream())
        self.k += 1
        return z[None]


vs = make_varstate(spec, th, sampler=StageSampler(normals))
stepper = V.FixedStepper(timeStep=1e-3, mode="Heun", maxStep=1e-2, increase_fac=1.3)
y_p, dt_p, info = stepper.step(0, V.TDVP(solver="eigh"), th.cuda(), evolutionEq=ep, psi=vs, nSamplesTDVP=N, nSamplesObs=N,
                               normFunction=lambda v, S: v @ S @ v, timings=None)
stage_metrics("Heun", y_p.cpu(), y_o, th, rec["S0"], rec["F0"], rec["ev"], rec["V"], dt)

# AdaptiveHeun (two attempts x five stages)
g = torch.Generator().manual_seed(5)
normals = [torch.randn(N, dim, generator=g, dtype=torch.float64) for _ in range(10)]
vo, to = O.VarStateOracle(spec, th), O.TDVPOracle()
calls = []


def f_a(y, t, k):
    z = O.latent_from_normals(spec, y, normals[len(calls)], None)
    calls.append(k)
    u, aux = to.rhs(y, t, vo, eo, z)
    if len(calls) == 6:        # first stage of the accepted attempt
        rec.update(S0=to.S0.clone(), F0=to.F0.clone(), ev=to.ev.clone(), V=to.V.clone())
    return u, aux


class R:
    pass


r = R()


def norm_rec(v, S):
    r.err = O.norm_fun(v, S)
    return r.err


O.adaptive_heun_step(f_a, th.clone(), 0.0, 2e-3, 1e30, 1.0, norm_rec, lambda: to.SExp)
tol = float(r.err) / 3.0
calls.clear()
y_o, real_o, next_o, _ = O.adaptive_heun_step(f_a, th.clone(), 0.0, 2e-3, tol, 1.0, O.norm_fun, lambda: to.SExp)
vs = make_varstate(spec, th, sampler=StageSampler(normals))
st = V.AdaptiveHeun(timeStep=2e-3, tol=tol, maxStep=1.0)
tp = V.TDVP(solver="eigh")
y_p, real_p, info = st.step(0.0, tp, th.cuda(), evolutionEq=ep, psi=vs, nSamplesTDVP=N, nSamplesObs=N, normFunction=lambda v, S: v @ S @ v, timings=None)
stage_metrics("AdaptiveHeun", y_p.cpu(), y_o, th, rec["S0"], rec["F0"], rec["ev"], rec["V"], real_o)
