"""Oracle parity of the BENCHED path on the BENCHED networks (BASELINE C2: d = 2, hidden (21,21), P = 4213; C3: hidden (43,43),
P = 16 181): one TDVP right-hand side through `solver="cholesky"` with bench.py's regularisation (diagonalShift 1e-4, svdTol 1e-6) --
which takes the pivoted low-rank route from 4096 parameters on, first as a planning call, then on its static plan with the previous
pivots as hints -- against the autodiff oracle's S0, F0 and the exact solve of the oracle's regularised matrix; the reference-default
`TDVP()` on the C2 net against the oracle's pseudo-inverse; the headline trajectory with a moving theta; and checkpoint -> resume at
the headline width.  Reference: tdvp.py:36-94.  Latent draws are injected into oracle and product."""
import os

import numpy as np
import pytest
import torch

from oracle import vmc_oracle as O
from helpers import make_spec, random_theta, make_varstate, FixedSampler, rel_err

pytestmark = pytest.mark.gpu

EQ = "advection_hamiltonian_wDiss"
SHIFT, SVD_TOL = 1e-4, 1e-6           # bench.py: DIAG_SHIFT, SVD_TOL


def _bench_net(hidden, N, seed, out_scale):
    """The bench.py network family (d = 2, 4 blocks, `hidden`, offset (1, 1), anharmonic Kramers operator lam = 0.1) with every
    parameter group active, `N` injected latent draws, oracle-side (Eloc, J, logp) and S0, F0, SExp of tdvp.py:36-47."""
    import vmc_pde_amd as V
    spec = make_spec(2, 4, hidden, latent="Gauss", coupling="no_add", offset=[1.0, 1.0], seed=seed)
    th = random_theta(spec, seed, out_scale=out_scale)
    g = torch.Generator().manual_seed(seed)
    z = O.latent_from_normals(spec, th, torch.randn(N, 2, generator=g, dtype=torch.float64))
    eo, ep = O.EvolutionEquationOracle(2, EQ), V.EvolutionEquation(dim=2, name=EQ)
    eo.eqParams[EQ]["params"]["lam"] = ep.eqParams[EQ]["params"]["lam"] = 0.1
    vo = O.VarStateOracle(spec, th)
    x, _ = vo.flow_sample(z)
    Eloc, J, logp = eo(vo, x, 0.1)
    return spec, th, z, ep, (Eloc, J, logp)


def _cholesky_route_parity(hidden, N, P_expect, seed, out_scale, device_solve):
    import vmc_pde_amd as V
    spec, th, z, ep, (Eloc, J, logp) = _bench_net(hidden, N, seed, out_scale)
    to = O.TDVPOracle(diagonalShift=SHIFT, svdTol=SVD_TOL)
    S_o, F_o, _ = to.get_tdvp_equation(Eloc, J, logp, N)                  # S0 + shift diag(S0) (tdvp.py:50-51), F0
    P = F_o.numel()
    assert P == P_expect
    # the system the regularised Cholesky solves: the reference's shifted matrix plus the absolute floor svdTol max diag(S0) I
    # (DESIGN.md section 5), built from the ORACLE's S0 and solved by a library LU
    A_o = S_o + SVD_TOL * torch.diag(to.S0).max() * torch.eye(P, dtype=torch.float64)
    if device_solve:                                                        # (P = 16 181: 3e12 flop -- on the GPU's library, not 16 host cores)
        exact = torch.linalg.solve(A_o.cuda(), F_o.cuda()).cpu()
    else:
        exact = torch.linalg.solve(A_o, F_o)
    nF = float(torch.linalg.norm(F_o))
    vs = make_varstate(spec, th, sampler=FixedSampler(z.cuda()))
    tp = V.TDVP(diagonalShift=SHIFT, svdTol=SVD_TOL, solver="cholesky")     # what bench.py constructs
    seen = []
    for call in range(3):                                                   # planning call, then two static calls (hints armed)
        u, _ = tp(th.cuda(), 0.1, vs, ep, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
        tp.check()
        seen.append(dict(tp.choleskyInfo))
        assert vs.numParameters == P
        assert rel_err(tp.S0, to.S0) < 1e-11 and rel_err(tp.F0, to.F0) < 1e-10
        assert abs(float(tp.ElocMean) - float(to.ElocMean)) < 1e-11 * (1 + abs(float(to.ElocMean)))
        du = u.cpu() - exact
        assert float(torch.linalg.norm(du) / torch.linalg.norm(exact)) < 1e-8, (call, float(torch.linalg.norm(du) / torch.linalg.norm(exact)))
        # ... and as the oracle's matrix sees the difference: |A_o (u - exact)| = |A_o u - F_o|, the residual of the PRODUCT's update in
        # the ORACLE's system
        assert float(torch.linalg.norm(A_o @ du)) < 1e-10 * nF, (call, float(torch.linalg.norm(A_o @ du)) / nF)
        assert float(tp.solverResidual) < 1e-11
        # tdvp.py:93 with the oracle's S0, F0 and Eloc
        err_o = float(1 + (exact @ to.S0 @ exact - 2 * to.F0 @ exact) / torch.mean(Eloc ** 2))
        assert abs(float(tp.tdvp_error) - err_o) < 1e-9
    assert [s["engine"] for s in seen] == ["lowrank"] * 3 and [s["mode"] for s in seen] == ["planning", "static", "static"], seen
    assert seen[2]["block_steps"] <= seen[1]["block_steps"] + 1 and seen[1]["planned_rank"] == seen[0]["rank"]
    return seen


def test_tdvp_rhs_parity_config2_net():
    seen = _cholesky_route_parity((21, 21), 512, 4213, seed=5, out_scale=0.1, device_solve=False)
    assert seen[0]["rank"] < 4213 // 6


def test_tdvp_rhs_parity_config3_net():
    seen = _cholesky_route_parity((43, 43), 256, 16181, seed=6, out_scale=0.05, device_solve=True)
    assert seen[0]["rank"] <= 768


def test_reference_default_tdvp_on_the_config2_net():
    """`TDVP()` exactly as the reference constructs it (tdvp.py:21-26) on the C2 network: the rank-revealing engine the default takes at this
    width against the oracle's host eigh + pseudo-inverse (tdvp.py:57-94)."""
    import vmc_pde_amd as V
    N = 512
    spec, th, z, ep, (Eloc, J, logp) = _bench_net((21, 21), N, seed=5, out_scale=0.1)
    to = O.TDVPOracle()
    upd_o, res_o, err_o = to.solve(Eloc, J, logp)
    vs = make_varstate(spec, th, sampler=FixedSampler(z.cuda()))
    tp = V.TDVP()
    upd_p, _ = tp(th.cuda(), 0.1, vs, ep, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
    tp.check()
    assert tp.spectralInfo["engine"] == "lowrank" and tp.spectralInfo["converged"], tp.spectralInfo
    P, nF, lmax = 4213, float(torch.linalg.norm(to.F0)), float(to.ev[-1])
    assert rel_err(tp.S0, to.S0) < 1e-11 and rel_err(tp.F0, to.F0) < 1e-10
    assert float((tp.ev.cpu() - to.ev).abs().max()) < 1e-10 * lmax
    du = upd_p.cpu() - upd_o
    assert float(torch.linalg.norm(to.S0 @ du)) < 1e-9 * nF
    well = to.ev > 1e-9 * lmax
    W = to.V[:, well]
    assert float(torch.linalg.norm(to.S0 @ (W @ (W.T @ du)))) < 1e-9 * nF
    msq = float((Eloc ** 2).mean())

    def restricted_error(u, Wm):
        uw = Wm @ (Wm.T @ u)
        return float(1 + (uw @ to.S0 @ uw - 2 * to.F0 @ uw) / msq)
    W7 = to.V[:, to.ev > 1e-7 * lmax]
    assert abs(restricted_error(upd_p.cpu(), W7) - restricted_error(upd_o, W7)) < 1e-10
    assert abs(float(tp.tdvp_error) - float(err_o)) < 1e-7
    # cluster invariants (eigenvalues chained by gaps below 1e-6 lambda_max): sum VtF^2 and sum rhoVar of every cluster above 1e-9
    bounds = [0] + [k for k in range(1, P) if float(to.ev[k] - to.ev[k - 1]) >= 1e-6 * lmax] + [P]
    f_p, f_o = (tp.VtF ** 2).cpu(), to.VtF ** 2
    rho_p, rho_o = tp.rhoVar.cpu(), to.rhoVar
    n = 0
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        if float(to.ev[hi - 1]) < 1e-9 * lmax:
            continue
        n += 1
        assert abs(float(f_p[lo:hi].sum()) - float(f_o[lo:hi].sum())) < 1e-8 * float(f_o[lo:hi].sum()) + 1e-12 * nF ** 2, (lo, hi)
        assert abs(float(rho_p[lo:hi].sum()) - float(rho_o[lo:hi].sum())) < 1e-8 * float(rho_o[lo:hi].sum()) + 1e-12 * float(rho_o.max()), (lo, hi)
    assert n >= 10


def test_small_svdtol_sends_the_spectral_route_to_the_dense_decomposition():
    """What the rank-revealing engine drops (trace of the remainder, a few 1e-13 max diag) is reported as exact zeros; the reference
    keeps every mode above 1e-14 lambda_max and weights it by 1 / (1 + (svdTol lambda_max / lambda)^6) (tdvp.py:82-85).  With the
    default svdTol = 1e-11 the dropped modes had weights below 1e-6; with a smaller svdTol they would count, so the engine is gated on
    it: same (ev, V, update) as the dense engine because it IS the dense engine then."""
    import vmc_pde_amd as V
    N = 2048
    spec, th, z, ep, _ = _bench_net((21, 21), 8, seed=5, out_scale=0.1)
    g = torch.Generator().manual_seed(5)
    z = O.latent_from_normals(spec, th, torch.randn(N, 2, generator=g, dtype=torch.float64))
    vs = make_varstate(spec, th, sampler=FixedSampler(z.cuda()))
    kw = dict(nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
    t_def = V.TDVP()
    t_def(th.cuda(), 0.0, vs, ep, **kw)
    info = t_def.spectralInfo
    assert info["engine"] == "lowrank" and info["trace_left"] <= 0.1 * 1e-11 * info["lambda_max_lower_bound"], info
    assert info["lambda_max_lower_bound"] <= float(t_def.ev[-1]) * (1 + 1e-12)          # a LOWER bound of lambda_max
    t_small, t_dense = V.TDVP(svdTol=1e-14), V.TDVP(svdTol=1e-14, spectralEngine="dense")
    u_s, _ = t_small(th.cuda(), 0.0, vs, ep, **kw)
    u_d, _ = t_dense(th.cuda(), 0.0, vs, ep, **kw)
    assert t_small.spectralInfo["engine"] == "dense" and t_small.spectralInfo.get("tried") == "lowrank", t_small.spectralInfo
    assert torch.equal(t_small.ev, t_dense.ev) and torch.equal(u_s, u_d)
    # a rejection by the svdTol gate cannot change from call to call: the attempt stays off (no factorisation wasted every 50 calls)
    # until svdTol is raised past the value that was rejected
    assert t_small._lowrank_gate_svdtol == 1e-14 and t_small._lowrank_backoff == 0
    for _ in range(2):
        t_small(th.cuda(), 0.0, vs, ep, **kw)
        assert t_small.spectralInfo == {"engine": "dense"}, t_small.spectralInfo          # not even tried
    t_small.svdTol = 1e-11
    t_small(th.cuda(), 0.0, vs, ep, **kw)
    assert t_small.spectralInfo["engine"] == "lowrank", t_small.spectralInfo


def _headline_objects(N, key=7):
    import vmc_pde_amd as V
    off = np.array([1.0, 1.0])
    sampler = V.Sampler(key=key, dim=2, name="Gauss", mcmc_info={"offset": off, "bound": 0.25})
    vs = V.VarState(sampler, 2, 1, 4, network_args={"intmediate": (21, 21), "offset": off, "latentSpaceName": "Gauss", "dim": 2})
    eq = V.EvolutionEquation(dim=2, name=EQ)
    return V, sampler, vs, eq


def test_headline_trajectory_stays_on_its_static_plan():
    """tools/run_headline_trajectory.py's statement as a test: 50 Heun steps (100 right-hand sides) of the headline network (harmonic Kramers
    problem, P = 4213) at N = 8192 on the reference's step schedule with a MOVING theta -- every solve after the first runs on a static
    plan of the low-rank route (re-planned only when the rank outgrows its pad) with at most 16 launched block steps, and at sampled steps its update equals the update of the blocked
    factorisation of the full matrix on the same samples to 1e-10 |F| as the matrix sees the difference."""
    N = 8192
    V, sampler, vs, eq = _headline_objects(N)
    _, sampler_d, vs_d, _ = _headline_objects(N)
    tp = V.TDVP(diagonalShift=SHIFT, svdTol=SVD_TOL, solver="cholesky")
    td = V.TDVP(diagonalShift=SHIFT, svdTol=SVD_TOL, solver="cholesky", choleskyEngine="dense")
    kw = dict(evolutionEq=eq, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
    modes, steps, ranks, worst = [], [], [], 0.0

    class Recording:
        """The stepper's right-hand side: the low-rank engine; at sampled calls the dense engine solves the same system on the same
        samples (a twin sampler set to the same stream position) and the two updates are compared."""

        def __init__(self):
            self.calls = 0

        def __call__(self, y, t, **rhs):
            nonlocal worst
            calls_before = sampler._calls
            u, info = tp(y, t, psi=vs, **{k: v for k, v in rhs.items() if k != "psi"})
            modes.append(tp.choleskyInfo.get("mode"))
            steps.append(tp.choleskyInfo.get("block_steps", 0))
            ranks.append(tp.choleskyInfo.get("rank", tp.choleskyInfo.get("planned_rank")))
            if self.calls % 10 == 3:
                sampler_d._calls = calls_before
                ud, _ = td(y, t, psi=vs_d, **{k: v for k, v in rhs.items() if k != "psi"})
                td.check()
                # same samples, same kernels: the same system up to the round-off of a different lagged shift (td's last call lies back)
                assert rel_err(td.S0, tp.S0) < 1e-12 and rel_err(td.F0, tp.F0) < 1e-11
                A = tp.S0 + torch.diag(SHIFT * torch.diag(tp.S0) + tp._floor)
                rel = float(torch.linalg.norm(A @ (u - ud)) / torch.linalg.norm(tp.F0))
                worst = max(worst, rel)
                assert rel < 1e-10, (self.calls, rel)
            self.calls += 1
            return u, info

        def check(self, *a):
            tp.check(*a)

    f = Recording()
    stepper = V.FixedStepper(timeStep=1e-4, mode="Heun", maxStep=1e-2, increase_fac=1.3)
    theta0 = vs.get_parameters().clone()
    t = 0.0
    for _ in range(50):
        y, dt, _ = stepper.step(t, f, vs.get_parameters(), psi=vs, **kw)
        vs.set_parameters(y)
        t += dt
    # every solve after the first runs on a static plan -- except where the numerical rank has outgrown the plan's pad (the random-init
    # network starts at rank ~63 with a pad of 128 and passes 96 within the first 50 calls at this sample count; the headline run at
    # N = 65 536 starts at 206 with a pad of 256 and never re-plans, profiles/r03_run7_headline_trajectory_f64.txt): a re-plan is
    # allowed only then, and at most once here
    planning = [i for i, m in enumerate(modes) if m != "static"]
    assert f.calls == 100 and planning[0] == 0 and len(planning) <= 2, planning
    for i in planning[1:]:
        pad_before = -(-(ranks[i - 1] + 40) // 128) * 128
        assert ranks[i] > pad_before - 32, (i, ranks[i - 1], ranks[i])
    assert worst < 1e-10
    assert max(steps) <= 16, max(steps)
    assert float(torch.linalg.norm(vs.get_parameters() - theta0)) > 1e-3 * float(torch.linalg.norm(theta0))     # theta moved
    assert torch.isfinite(vs.get_parameters()).all() and t > 0.3


def test_resume_is_bit_identical_on_the_static_lowrank_route(tmp_path):
    """TDVP.state_dict carries what the low-rank route inherits (plan, pivot list, step count) besides the lagged means: six Heun steps of
    the headline network (P = 4213, N = 4096), against three steps + checkpoint through the host + three steps with fresh objects."""
    N = 4096
    V, _, _, _ = _headline_objects(N)

    def run(n_steps, start=None):
        _, sampler, vs, eq = _headline_objects(N)
        tp = V.TDVP(diagonalShift=SHIFT, svdTol=SVD_TOL, solver="cholesky")
        stepper = V.FixedStepper(timeStep=1e-4, mode="Heun", maxStep=1e-2, increase_fac=1.3)
        t = 0.0
        if start is not None:
            st = torch.load(start, weights_only=False)
            vs.set_parameters(st["theta"].cuda())
            t, stepper.dt, sampler._calls = st["t"], st["dt"], st["sampler_calls"]
            tp.load_state_dict(st["tdvp"])
        modes = []
        for _ in range(n_steps):
            y, dt, _ = stepper.step(t, tp, vs.get_parameters(), evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N, timings=None,
                                    observables=False)
            vs.set_parameters(y)
            t += dt
            modes.append(tp.choleskyInfo["mode"])
        tp.check()
        return vs, tp, stepper, sampler, t, modes

    vs6, tp6, _, _, t6, modes6 = run(6)
    vs3, tp3, stepper3, sampler3, t3, _ = run(3)
    ckpt = os.path.join(str(tmp_path), "ckpt.pt")
    torch.save({"theta": vs3.get_parameters().cpu(), "t": t3, "dt": stepper3.dt, "sampler_calls": sampler3._calls, "tdvp": tp3.state_dict()}, ckpt)
    del vs3, tp3
    vsr, tpr, _, _, tr, modes_r = run(3, start=ckpt)
    assert modes6 == ["static"] * 6 and modes_r == ["static"] * 3           # the resumed run never re-plans: it continues the plan
    assert tr == t6
    assert torch.equal(vsr.get_parameters(), vs6.get_parameters())
    assert torch.equal(tpr.F0, tp6.F0) and float(tpr.solverResidual) == float(tp6.solverResidual)
    assert tpr.choleskyInfo == tp6.choleskyInfo
