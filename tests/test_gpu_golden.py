"""HIP path against the committed oracle vectors (tests/golden/oracle_vectors.npz, oracle_heun_step.npz).
Nothing here imports the oracle: the fixtures are the target.  Tolerances: SURVEY.md 8(d)."""
import os

import numpy as np
import pytest
import torch

from helpers_product import FixedSampler, build_varstate, rel_err

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "oracle_vectors.npz"))
T_EVAL = 0.3
FULL_MATRIX_MAX_P = 120
# same table as tests/golden/make_oracle_fixtures.py (kept literal so that this file does not import the oracle)
CASES = [
    ("d2_b4_h1_gauss_noadd", 2, 4, (1,), "Gauss", "no_add", ["diffusion", "diffusion_drift", "advection_hamiltonian_wDiss"]),
    ("d2_b1_h44_gauss_plain", 2, 1, (4, 4), "Gauss", "plain", ["advection_paper", "advection_hamiltonian"]),
    ("d3_b4_h3_gauss_jaceq1", 3, 4, (3,), "Gauss", "jac_eq_1", ["diffusion", "diffusion_anisotropic"]),
    ("d6_b4_h3_gauss_diffadd", 6, 4, (3,), "Gauss", "different_add", ["advection_hamiltonian_wDiss", "diffusion_anisotropic"]),
    ("d8_b1_h3_student_noadd", 8, 1, (3,), "Student_t", "no_add", ["diffusion"]),
    ("d8_b4_h44_student_noadd", 8, 4, (4, 4), "Student_t", "no_add", ["diffusion", "diffusion_drift"]),
    ("d3_b2_h3_gauss_noadd_global", 3, 2, (3,), "Gauss", "no_add+global", ["diffusion"]),      # global_change, net.py:72
]


def _t(a):
    return torch.as_tensor(np.asarray(a)).cuda()


def _probe(P):
    return torch.cos(0.37 * torch.arange(P, dtype=torch.float64) + 0.1)


def _equation(V, dim, name, tag):
    ep = V.EvolutionEquation(dim=dim, name=name)
    if "hamiltonian" in name:
        ep.eqParams[name]["params"]["lam"] = 0.1
    if name == "diffusion_anisotropic":
        ep.eqParams[name]["D"] = torch.as_tensor(GOLD[f"{tag}/{name}/D"])
    return ep


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_flow_and_rhs_against_committed_vectors(case):
    import vmc_pde_amd as V
    tag, dim, depth, hidden, latent, coupling, eqs = case
    th, z = _t(GOLD[f"{tag}/theta"]), _t(GOLD[f"{tag}/z"])
    vs = build_varstate(dim, depth, hidden, latent, coupling, GOLD[f"{tag}/offset"], GOLD[f"{tag}/inds_up"], GOLD[f"{tag}/inds_down"],
                        th, sampler=FixedSampler(z))
    P = vs.numParameters
    assert P == th.numel()
    x, logp_s = vs.flow_sample(z)
    assert rel_err(x, GOLD[f"{tag}/x"]) < 1e-12 and rel_err(logp_s, GOLD[f"{tag}/logp_sample"]) < 1e-12
    xg = _t(GOLD[f"{tag}/x"])
    logp, g, J = vs(xg, mode="eval_coordgrads")
    assert rel_err(logp, GOLD[f"{tag}/logp"]) < 1e-12
    assert rel_err(g, GOLD[f"{tag}/grad_x"]) < 1e-12
    Jg = GOLD[f"{tag}/J"]
    assert float((J.cpu() - torch.as_tensor(Jg)).abs().max()) < 1e-12 * np.abs(Jg).max()
    assert rel_err(vs.hessian(xg), GOLD[f"{tag}/hessian"]) < 1e-11
    for name in eqs:
        pre = f"{tag}/{name}"
        ep = _equation(V, dim, name, tag)
        Eloc, _, _ = ep(vs, xg, T_EVAL)
        assert rel_err(Eloc, GOLD[f"{pre}/Eloc"]) < 1e-11, name
        tp = V.TDVP(solver="eigh")
        upd, _ = tp(th, T_EVAL, vs, ep, nSamplesTDVP=z.shape[1], nSamplesObs=z.shape[1], timings=None)
        assert rel_err(tp.F0, GOLD[f"{pre}/F0"]) < 1e-10
        st = GOLD[f"{pre}/Eloc_stats"]
        assert abs(float(tp.ElocMean) - st[0]) < 1e-10 * (1 + abs(st[0])) and abs(float(tp.ElocVar) - st[1]) < 1e-9 * (1 + st[1])
        for key in ("S0", "SExp"):
            Mx = getattr(tp, key).cpu()
            if P <= FULL_MATRIX_MAX_P:
                assert rel_err(Mx, GOLD[f"{pre}/{key}"]) < 1e-11, (name, key)
            else:
                assert rel_err(torch.diag(Mx), GOLD[f"{pre}/{key}_diag"]) < 1e-11, (name, key)
                assert rel_err(Mx @ _probe(P), GOLD[f"{pre}/{key}_probe"]) < 1e-11, (name, key)
        evg = GOLD[f"{pre}/ev"]
        big = evg > 1e-9 * evg[-1]
        assert np.allclose(tp.ev.cpu().numpy()[big], evg[big], rtol=1e-6, atol=0)
        # S is rank deficient: the update is compared in the S-norm
        du = upd.cpu() - torch.as_tensor(GOLD[f"{pre}/update"])
        assert float(torch.linalg.norm(tp.S0.cpu() @ du)) < 1e-9 * np.linalg.norm(GOLD[f"{pre}/F0"]), name
        # ... and, with the TDVP error, restricted to the modes with lambda > 1e-9 lambda_max at SURVEY's 1e-9 / 1e-10
        Vw = tp.V.cpu()[:, torch.as_tensor(tp.ev.cpu().numpy() > 1e-9 * float(tp.ev[-1]))]
        S0c, F0c = tp.S0.cpu(), tp.F0.cpu()
        msq = float(tp.ElocVar) + float(tp.ElocMean) ** 2

        def restricted(u):
            uw = Vw @ (Vw.T @ u)
            return uw, float(1 + (uw @ S0c @ uw - 2 * F0c @ uw) / msq)

        (uw_p, err_p), (uw_g, err_g) = restricted(upd.cpu()), restricted(torch.as_tensor(GOLD[f"{pre}/update"]))
        assert float(torch.linalg.norm(S0c @ (uw_p - uw_g))) < 1e-9 * np.linalg.norm(GOLD[f"{pre}/F0"]), name
        assert abs(err_p - err_g) < 1e-10, name
        # tdvp_error = 1 - F^T S^+ F / <E^2> feels the eigenvalues that sit on the soft cut-off (round-off level, H2)
        assert abs(float(tp.tdvp_error) - float(GOLD[f"{pre}/tdvp_error"])) < 1e-6, name


def test_heun_step_against_committed_vector():
    """One FixedStepper('Heun') step (stepper.py:129-139) with the stage normals of the fixture."""
    import vmc_pde_amd as V
    from vmc_pde_amd import _native as nat
    h = np.load(os.path.join(HERE, "golden", "oracle_heun_step.npz"))
    dim, N = 2, h["normals0"].shape[0]
    normals = [_t(h["normals0"]), _t(h["normals1"])]
    zs = []

    class StageSampler:
        key = 1

        def __init__(self):
            self.k = 0

        def __call__(self, numSamples, dist_params, multipleOf=1):
            z = torch.empty((N, dim), dtype=torch.float64, device="cuda")
            nat.call("vmcpde_latent_from_normals", dist_params["_model"].handle, nat.ptr(dist_params["_theta"]),
                     nat.ptr(normals[self.k]), None, N, nat.ptr(z), nat.stream())
            zs.append(z.clone())
            self.k += 1
            return z[None]

    th0 = _t(h["theta0"])
    vs = build_varstate(dim, 4, (1,), "Gauss", "no_add", h["offset"], h["inds_up"], h["inds_down"], th0, sampler=StageSampler())
    stepper = V.FixedStepper(timeStep=1e-5, mode="Heun", maxStep=1e-4, increase_fac=1.3)
    engine, slopes = V.TDVP(solver="eigh"), []

    def rhs(y_, t_, **kw):                       # records the slope of every stage
        u_, info_ = engine(y_, t_, **kw)
        slopes.append(u_.clone())
        return u_, info_

    y, dt, info = stepper.step(0, rhs, th0, evolutionEq=V.EvolutionEquation(dim=dim, name="diffusion"), psi=vs,
                               nSamplesTDVP=N, nSamplesObs=N, normFunction=lambda v, S: v @ S @ v, timings=None)
    assert dt == float(h["dt"])
    # stage 0 sees identical inputs; stage 1 is evaluated at theta0 + dt*k0, and k0 is only defined up to the near-null
    # modes of the rank-deficient S (pseudo-inverse cut-off 1e-11), which is what the looser bounds reflect
    assert rel_err(zs[0], h["z0"][0]) < 1e-13 and rel_err(zs[1], h["z1"][0]) < 1e-6
    step = torch.as_tensor(h["theta1"] - h["theta0"])
    err = float(torch.linalg.norm((y.cpu() - torch.as_tensor(h["theta0"])) - step) / torch.linalg.norm(step))
    print("heun step relative deviation", err)
    assert err < 1e-4
    # The basis-independent statement, stage by stage: at the COMMITTED state of each stage (theta0, and the oracle's predictor
    # theta0 + dt k0) the product's update agrees with the committed one in the S-norm of that stage at 1e-9 |F|, and
    # projected on the modes above 1e-9 lambda_max.  (The composite step cannot be held to that: the second stage is evaluated at a
    # state that carries the first stage's null-space freedom, and S, F depend on it through second derivatives of log p --
    # measured 1e-7 |F| dt in the S-norm, tools/diag/invariant_tolerances.py.)
    for k in range(2):
        smp = StageSampler()
        smp.k = k
        vk = build_varstate(dim, 4, (1,), "Gauss", "no_add", h["offset"], h["inds_up"], h["inds_down"], _t(h["stage_theta"][k]), sampler=smp)
        tp = V.TDVP(solver="eigh")
        u, _ = tp(_t(h["stage_theta"][k]), 0.0, vk, V.EvolutionEquation(dim=dim, name="diffusion"), nSamplesTDVP=N, nSamplesObs=N, timings=None,
                  integrals=False)
        du = u.cpu() - torch.as_tensor(h["stage_update"][k])
        S0, nF = tp.S0.cpu(), float(np.linalg.norm(h["stage_F"][k]))
        assert rel_err(tp.F0, h["stage_F"][k]) < 1e-10
        assert float(torch.linalg.norm(S0 @ du)) < 1e-9 * nF, (k, float(torch.linalg.norm(S0 @ du)) / nF)
        ev, Vv = tp.ev.cpu(), tp.V.cpu()
        W = Vv[:, ev > 1e-9 * ev[-1]]
        assert float(torch.linalg.norm(S0 @ (W @ (W.T @ du)))) < 1e-9 * nF, k
    # and the step itself is the explicit trapezoid of the two slopes the product computed (stepper.py:135-139), to rounding
    assert len(slopes) == 2
    assert float((y - (th0 + 0.5 * dt * (slopes[0] + slopes[1]))).abs().max()) <= 4e-16 * float(th0.abs().max() + dt * slopes[0].abs().max())
