"""GPU parity: the flow kernels of csrc/flow.hip and csrc/sampler.hip (through the C ABI / host classes) against the
autodiff CPU oracle on identical seeded inputs.  Tolerances are the fp64 figures of SURVEY.md 8(d)."""
import numpy as np
import pytest
import torch

from oracle import vmc_oracle as O
from helpers import make_spec, random_theta, make_varstate, rel_err

pytestmark = pytest.mark.gpu

CONFIGS = [
    # dim, depth, hidden, latent, coupling
    (2, 4, (1,), "Gauss", "no_add"),            # the 'mwe' net of main.py:38,73
    (2, 1, (3,), "Gauss", "no_add"),
    (3, 4, (3,), "Gauss", "no_add"),            # odd d
    (6, 4, (3,), "Gauss", "different_add"),     # the published phase-space net (P = 411)
    (8, 1, (4, 4), "Student_t", "no_add"),
    (8, 3, (6,), "Student_t", "no_add"),        # the published d=8 Student-t net (P = 393)
    (6, 4, (4, 4), "Gauss", "plain"),
    (4, 2, (3,), "Student_t", "jac_eq_1"),
    (2, 4, (21, 21), "Gauss", "no_add"),        # BASELINE C2 net (P = 4213)
    (2, 2, (9, 9, 9), "Gauss", "different_add"),
    (3, 3, (4,), "Gauss", "no_add+global"),              # global_change (net.py:72,80-82,115-116,150-151), odd d
    (6, 11, (3,), "Student_t", "different_add+global"),  # ... with t-networks; 11 blocks: 'blocks_10' sorts before 'blocks_2'
    (16, 2, (5, 4, 3), "Student_t", "different_add"),    # the ABI's maxima: d = VMCPDE_MAX_DIM (136 Hessian directions), three hidden layers
    (2, 2, (50,), "Gauss", "no_add"),                    # wider than 8 waves x 6 outputs: two Taylor buffers, eps in an LDS line, 32-sample workgroups
    (4, 1, (19, 19), "Gauss", "different_add"),          # the in-place plan with the s stash and both networks of a half-step
]


def _setup(cfg, n, seed=11):
    dim, depth, hidden, latent, coupling = cfg
    off = np.linspace(-0.5, 0.7, dim)
    spec = make_spec(dim, depth, hidden, latent=latent, coupling=coupling, offset=off, seed=seed)
    th = random_theta(spec, seed)
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(1, n, dim, generator=g, dtype=torch.float64) * 0.8 + torch.as_tensor(off)
    return spec, th, z


@pytest.mark.parametrize("cfg", CONFIGS)
def test_parameter_layout_matches_oracle(cfg):
    spec, th, _ = _setup(cfg, 4)
    vs = make_varstate(spec, th)
    assert vs.numParameters == O.num_parameters(spec)
    lay_o = [(n, tuple(s)) for n, s in O.param_layout(spec)]
    lay_p = [("/".join(p), tuple(s)) for p, _, s in vs.net.layout]
    assert lay_o == lay_p
    assert vs.paramShapes == [(int(np.prod(s)), tuple(s)) for _, s in lay_o]
    # set/get round trip and nested views (main.py:178 reads params["params"]["dist_params"])
    assert torch.equal(vs.get_parameters().cpu(), th)
    tree = O.unflatten(spec, th)
    assert torch.equal(vs.params["params"]["mu"].cpu(), tree["mu"])
    assert torch.equal(vs.params["params"]["myINN"]["blocks_0"]["s1"]["Dense_0"]["kernel"].cpu(), tree["myINN/blocks_0/s1/Dense_0/kernel"])


@pytest.mark.parametrize("cfg", CONFIGS)
def test_flow_sample_and_eval_parity(cfg):
    n = 70 if cfg[2] != (21, 21) else 40        # ragged against the 64-lane wave on purpose
    spec, th, z = _setup(cfg, n)
    vo = O.VarStateOracle(spec, th)
    vs = make_varstate(spec, th)
    # J5: z -> (x, logp)
    x_o, lp_o = vo.flow_sample(z)
    x_p, lp_p = vs.flow_sample(z)
    assert rel_err(x_p, x_o) < 1e-12
    assert rel_err(lp_p, lp_o) < 1e-12
    # forward direction (main.py:81)
    y_p, l_p = vs.net.apply(vs.params, x_o[0, 0].clone(), evaluate=False, inv=False)
    y_o, l_o = O.inn_w_prob(spec, th, x_o[0, 0], evaluate=False, inv=False)
    assert rel_err(y_p, y_o) < 1e-12 and abs(float(l_p) - float(l_o)) < 1e-11 * max(1, abs(float(l_o)))
    # evaluate=True with inv=True (net.py:209-213 passes `inv` through; no caller of the reference uses it): single point and batch
    e_p = vs.net.apply(vs.params, z[0, 1].clone(), evaluate=True, inv=True)
    e_o = O.inn_w_prob(spec, th, z[0, 1], evaluate=True, inv=True)
    assert abs(float(e_p) - float(e_o)) < 1e-11 * max(1, abs(float(e_o)))
    e_b = vs.net.apply(vs.params, z[0, :5].clone(), evaluate=True, inv=True)
    assert e_b.shape == (5,) and abs(float(e_b[1]) - float(e_o)) < 1e-11 * max(1, abs(float(e_o)))
    # J1, J3, J6
    lo, go, Jo = vo(x_o, "eval_coordgrads")
    lp, gp, Jp = vs(x_o, mode="eval_coordgrads")
    assert lp.shape == (1, n) and gp.shape == (1, n, spec.dim) and Jp.shape == (1, n, vs.numParameters)
    assert rel_err(lp, lo) < 1e-12
    assert rel_err(gp, go) < 1e-11
    assert float((Jp.cpu() - Jo).abs().max()) < 1e-11 * float(Jo.abs().max())
    assert rel_err(vs(x_o, mode="eval"), lo) < 1e-12
    # J4
    Ho = vo.hessian(x_o)
    Hp = vs.hessian(x_o)
    assert float((Hp.cpu() - Ho).abs().max()) < 1e-10 * float(Ho.abs().max())


@pytest.mark.parametrize("cfg", [CONFIGS[3], CONFIGS[8], CONFIGS[11]])
def test_activation_workspace_variant_gives_identical_results(cfg):
    """vmcpde_flow_eval with a workspace (the forward pass keeps the activations, the reverse sweep reads them back) against the
    default (re-evaluation).  The forward results (logp, directional second derivatives) are the same code: bit for bit.  The
    reverse sweep of the default walks the state backwards through the exact inverse of every coupling step (x = y e^{-s}), which
    reproduces the forward activations only to rounding, so grad_x and the Jacobian agree to ~1e-15, not bitwise."""
    spec, th, z = _setup(cfg, 150)
    out = {}
    for keep in (False, True):
        vs = make_varstate(spec, th)
        vs.keep_activations = keep
        x, _ = vs.flow_sample(z)
        r = vs.evaluate(x, dirs=np.eye(spec.dim)[:2])
        out[keep] = (r["logp"].clone(), r["g"].clone(), r["J"].clone(), r["hv"].clone())
    assert torch.equal(out[False][0], out[True][0]) and torch.equal(out[False][3], out[True][3])
    assert rel_err(out[True][1], out[False][1]) < 1e-13
    assert float((out[True][2] - out[False][2]).abs().max()) < 1e-13 * float(out[False][2].abs().max())


def test_empty_and_single_sample():
    spec, th, z = _setup(CONFIGS[2], 1)
    vs = make_varstate(spec, th)
    x, lp = vs.flow_sample(z)
    assert x.shape == (1, 1, 3)
    from vmc_pde_amd import _native as nat
    x0 = torch.empty((0, 3), dtype=torch.float64, device=vs.device)
    th_d = vs.get_parameters()
    nat.call("vmcpde_flow_eval", vs.net.handle, nat.ptr(th_d), nat.ptr(x0), 0, None, 0, None, None, None, None, 0, 0, None, 0, nat.stream())   # empty batch: no-op


@pytest.mark.parametrize("latent", ["Gauss", "Student_t"])
def test_latent_from_normals_parity(latent):
    from vmc_pde_amd import _native as nat
    spec, th, _ = _setup((6, 2, (3,), latent, "no_add"), 4)
    vs = make_varstate(spec, th)
    g = torch.Generator().manual_seed(5)
    n = 257
    nrm = torch.randn(n, 6, generator=g, dtype=torch.float64)
    chi2 = torch.distributions.Chi2(torch.tensor(3.0, dtype=torch.float64)).sample((n,))
    z_o = O.latent_from_normals(spec, th, nrm, chi2)
    z = torch.empty((n, 6), dtype=torch.float64, device=vs.device)
    th_d, nrm_d, chi2_d = vs.get_parameters(), nrm.to(vs.device), chi2.to(vs.device)   # keep the buffers alive across the launch
    nat.call("vmcpde_latent_from_normals", vs.net.handle, nat.ptr(th_d), nat.ptr(nrm_d), nat.ptr(chi2_d), n, nat.ptr(z), nat.stream())
    assert rel_err(z, z_o[0]) < 1e-13


@pytest.mark.parametrize("latent", ["Gauss", "Student_t"])
def test_exact_sampler_statistics(latent):
    # RNG streams cannot match JAX (H4): pin the law instead.  Gauss: mean mu+offset, cov L L^T; t: cov nu/(nu-2) L L^T
    spec, th, _ = _setup((3, 2, (3,), latent, "no_add"), 4)
    if latent == "Student_t":
        off_d = [i for i, (nme, s) in enumerate(O.param_layout(spec)) if nme == "dist_params"]
        th = th.clone()
        names = [nme for nme, s in O.param_layout(spec) for _ in range(int(np.prod(s)))]
        th[names.index("dist_params")] = np.log(7.0)     # nu = 8
    vs = make_varstate(spec, th)
    N = 400000
    p = vs.params["params"]
    from vmc_pde_amd import util
    dist = {"S": util.build_cov_matrix(p["L"], p["L_diag"], 3), "mu": p["mu"], "dist_params": p["dist_params"], "_model": vs.net,
            "_theta": vs.get_parameters()}
    z = vs.sampler(N, dist)[0].cpu()
    tree = O.unflatten(spec, th)
    S = O.build_cov_matrix(tree["L"], tree["L_diag"], 3)
    scale = 1.0 if latent == "Gauss" else 8.0 / 6.0
    mean = tree["mu"] + torch.as_tensor(spec.offset)
    assert float((z.mean(0) - mean).abs().max()) < 0.012
    C = torch.cov(z.T)
    assert float((C - scale * S).abs().max()) < 0.03 * float(S.abs().max()) * scale
    # successive calls use fresh streams; same key + same call index reproduces
    z2 = vs.sampler(1000, dist)[0]
    assert not torch.equal(z2.cpu(), z[:1000])
    # sampler.py:86 adds the SAMPLER's offset (mcmc_info), which a caller may choose different from the network's
    import vmc_pde_amd as V
    s2 = V.Sampler(key=1, dim=3, name=latent, mcmc_info={"offset": np.asarray(spec.offset) + np.array([0.5, -1.0, 2.0]), "bound": 0.25})
    za, zb = s2(1000, dist)[0].cpu(), V.Sampler(key=1, dim=3, name=latent, mcmc_info={"offset": np.asarray(spec.offset), "bound": 0.25})(1000, dist)[0].cpu()
    assert float((za - zb - torch.tensor([0.5, -1.0, 2.0], dtype=torch.float64)).abs().max()) < 1e-14


def test_sample_sharding_is_world_size_invariant():
    # sample n depends only on (key, call, global index): rank shards concatenate to the single-rank draw
    from vmc_pde_amd import _native as nat
    spec, th, _ = _setup((2, 2, (3,), "Gauss", "no_add"), 4)
    vs = make_varstate(spec, th)
    full = torch.empty((1000, 2), dtype=torch.float64, device=vs.device)
    a = torch.empty((400, 2), dtype=torch.float64, device=vs.device)
    b = torch.empty((600, 2), dtype=torch.float64, device=vs.device)
    t = vs.get_parameters()
    nat.call("vmcpde_sample_latent", vs.net.handle, nat.ptr(t), 9, 3, 0, 1000, nat.ptr(full), nat.stream())
    nat.call("vmcpde_sample_latent", vs.net.handle, nat.ptr(t), 9, 3, 0, 400, nat.ptr(a), nat.stream())
    nat.call("vmcpde_sample_latent", vs.net.handle, nat.ptr(t), 9, 3, 400, 600, nat.ptr(b), nat.stream())
    assert torch.equal(torch.cat([a, b]), full)


def test_mcmc_parity_with_injected_uniforms():
    import ctypes as C
    from vmc_pde_amd import _native as nat
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(1)
    chains, steps = 30, 200       # numChains=30 as main.py:69
    off = np.array([0.25, 0.25])
    st0 = O.radial_update_prop(rng.random(chains), rng.random(chains), 0.25, off)
    ur, up, ua = rng.random((steps, chains)), rng.random((steps, chains)), rng.random((steps, chains))
    out_o, nacc_o = O.mcmc_chain(st0, ur, up, ua, 0.25, off)
    states = torch.as_tensor(st0).to(dev).contiguous()
    out = torch.empty((steps * chains, 2), dtype=torch.float64, device=dev)
    nacc = torch.zeros(1, dtype=torch.int64, device=dev)
    off2 = (C.c_double * 2)(*off.tolist())
    ur_d, up_d, ua_d = (torch.as_tensor(a).to(dev) for a in (ur, up, ua))   # keep the buffers alive across the launch
    nat.call("vmcpde_mcmc_from_uniforms", nat.ptr(ur_d), nat.ptr(up_d), nat.ptr(ua_d), chains, steps, 0.25, off2, nat.ptr(states),
             nat.ptr(out), nat.ptr(nacc), nat.stream())
    assert int(nacc.item()) == nacc_o
    assert float(np.abs(out.cpu().numpy() - out_o).max()) < 1e-14


def test_mcmc_sampler_stationary_density():
    import vmc_pde_amd as V
    off = np.array([0.25, 0.25])
    s = V.Sampler(key=3, numChains=64, dim=2, name="cos_dist", mcmc_info={"offset": off, "bound": 0.25})
    x = s(64 * 3000, None)[0].cpu().numpy()
    assert x.shape == (64 * 3000, 2)
    r = 4 * np.linalg.norm(x - off, axis=1)
    rr = np.linspace(0, 1, 20001)
    w = (1 + np.cos(np.pi * rr)) * rr
    expect = np.trapezoid(w * rr, rr) / np.trapezoid(w, rr)
    assert abs(r.mean() - expect) < 0.01 and r.max() <= 1.0 + 1e-12


EQS = ["diffusion", "diffusion_drift", "diffusion_anisotropic", "advection_hamiltonian", "advection_hamiltonian_wDiss",
       "advection_hamiltonian+ring", "advection_hamiltonian_wDiss+ring"]


@pytest.mark.parametrize("name", EQS)
def test_rhs_parity(name):
    """"+ring": the coupled branch of the Hamiltonian (evolutionEq.py:33-38), which the reference defines and never selects."""
    import vmc_pde_amd as V
    name, _, ring = name.partition("+")
    cfg = (6, 2, (3,), "Gauss", "different_add")
    spec, th, z = _setup(cfg, 50)
    vo = O.VarStateOracle(spec, th)
    vs = make_varstate(spec, th)
    x, _ = vo.flow_sample(z)
    eo = O.EvolutionEquationOracle(6, name)
    ep = V.EvolutionEquation(dim=6, name=name)
    if name == "diffusion_anisotropic":
        ep.eqParams[name]["D"] = eo.eqParams[name]["D"].clone()
    if "hamiltonian" in name:
        eo.eqParams[name]["params"]["lam"] = 0.1      # BASELINE C2: anharmonic drift
        ep.eqParams[name]["params"]["lam"] = 0.1
        if ring:
            eo.eqParams[name]["params"]["coupled"] = True
            ep.eqParams[name]["params"]["coupled"] = True
    E_o, J_o, lp_o = eo(vo, x, 0.3)
    E_p, J_p, lp_p = ep(vs, x, 0.3)
    assert E_p.shape == (1, 50)
    assert rel_err(E_p, E_o) < 1e-10
    assert rel_err(lp_p, lp_o) < 1e-12
    assert float((J_p.cpu() - J_o).abs().max()) < 1e-11 * float(J_o.abs().max())


def test_rhs_advection_paper_time_dependent():
    import vmc_pde_amd as V
    spec, th, z = _setup((2, 3, (3,), "Gauss", "no_add"), 33)
    vo, vs = O.VarStateOracle(spec, th), make_varstate(spec, th)
    x, _ = vo.flow_sample(z)
    for t in (0.0, 1.7):
        E_o, _, _ = O.EvolutionEquationOracle(2, "advection_paper")(vo, x, t)
        E_p, _, _ = V.EvolutionEquation(dim=2, name="advection_paper")(vs, x, t)
        assert rel_err(E_p, E_o) < 1e-11


def test_ka3_density_is_normalised_on_the_reference_grid():
    # main.py:100-105 / var_state.py:88-91: integral of exp(logp) over the reference's regular grid (same spacing 0.1), for an active flow (every coupling
    # variant changes the density, none may change its mass)
    import vmc_pde_amd as V
    from vmc_pde_amd import grid as G
    for coupling in ("no_add", "different_add", "jac_eq_1", "plain"):
        spec = make_spec(2, 3, (4,), coupling=coupling, offset=[0.3, -0.2], seed=5)
        th = random_theta(spec, 5, out_scale=0.02, latent_scale=0.2)
        vs = make_varstate(spec, th)
        grid = G.Grid(np.array([30.0, 30.0]), 600, sym=True)     # the multiplicative variants spread mass beyond +-10
        assert abs(float(vs.integrate(grid)) - 1.0) < 1e-5, coupling


@pytest.mark.parametrize("kw,msg", [(dict(dim=17), "dim 17"), (dict(depth=17), "depth 17"), (dict(hidden=(3, 3, 3, 3)), "n_hidden 4"),
                                    (dict(overlap=True), "bad ind_down")])
def test_model_descriptor_limits_fail_loudly(kw, msg):
    """vmcpde_create rejects what the ABI cannot represent (VMCPDE_MAX_DIM / _DEPTH / _HIDDEN_LAYERS) and partitions that are not a
    split of the coordinates (var_state.py:116-117) with an error naming the field -- no silent truncation."""
    import vmc_pde_amd as V
    dim, depth, hidden = kw.get("dim", 4), kw.get("depth", 2), kw.get("hidden", (3,))
    ups = [list(range(dim // 2)) for _ in range(depth)]
    dns = [list(range(dim // 2, dim)) for _ in range(depth)]
    if kw.get("overlap"):
        dns[0][0] = ups[0][0]
    off = np.zeros(dim)
    sampler = V.Sampler(key=1, dim=dim, name="Gauss", mcmc_info={"offset": off, "bound": 0.25})
    with pytest.raises((RuntimeError, ValueError, AssertionError)) as err:
        V.VarState(sampler, dim, 1, depth, network_args=dict(intmediate=hidden, offset=off, latentSpaceName="Gauss", dim=dim),
                   inds_up=ups, inds_down=dns)
    assert msg in str(err.value), str(err.value)


@pytest.mark.parametrize("cfg", CONFIGS)
@pytest.mark.parametrize("jdtype", [None, "f32"])
def test_evaluate_from_latent_matches_sample_then_evaluate(cfg, jdtype):
    """ABI 7, vmcpde_flow_eval_from_latent: one launch from the LATENT points -- the reverse sweep starts at z and produces x on its way, the Taylor
    pass runs forward from that x -- against the two launches it replaces (inverse flow, then the evaluation at its x) and against the oracle's
    inverse flow: x, log p, grad_x, every directional second derivative (more directions than one pass carries where d allows) and the Jacobian.
    Ragged sample counts: idle and shadow lanes take part in the barriers and must not disturb their neighbours."""
    dim = cfg[0]
    for n in ((70, 1) if cfg[2] != (21, 21) else (40,)):
        spec, th, z = _setup(cfg, n)
        kw = {} if jdtype is None else {"jacobian_dtype": torch.float32}
        vs, vs2 = make_varstate(spec, th, **kw), make_varstate(spec, th, **kw)
        if cfg[4].endswith("+global"):
            # net.py:150-151: the reference's inverse direction with global_change is not the inverse of its forward map (restated as written in
            # the oracle and in vmcpde_flow_sample), so the sweep's x would not be the reference's samples: the entry point refuses such a model
            from vmc_pde_amd import _native as nat
            with pytest.raises(nat.VmcpdeError, match="global_change"):
                vs2.evaluate_from_latent(z, dirs=np.eye(dim))
            return
        dirs = np.eye(dim) if dim <= 8 else np.eye(dim)[::3]
        x1, _ = vs.flow_sample(z)
        r1 = vs.evaluate(x1, dirs=dirs)
        ref = {k: r1[k].clone() for k in ("logp", "g", "hv")}
        J1 = r1["J"].clone()
        r2 = vs2.evaluate_from_latent(z, dirs=dirs)
        x_o, _ = O.VarStateOracle(spec, th).flow_sample(z)
        assert rel_err(r2["x"], x_o[0]) < 1e-12 and rel_err(r2["x"], x1[0]) < 1e-13
        assert rel_err(r2["logp"], ref["logp"]) < 1e-12 and rel_err(r2["g"], ref["g"]) < 1e-10 and rel_err(r2["hv"], ref["hv"]) < 1e-9
        tolJ = 1e-11 if jdtype is None else 1e-6
        assert float((r2["J"].double() - J1.double()).abs().max()) < tolJ * float(J1.double().abs().max())
        assert torch.isfinite(r2["J"]).all() and r2["J"].shape == J1.shape


def test_tdvp_call_with_and_without_the_fused_sample_evaluation(monkeypatch):
    """TDVP.__call__ draws the latent points and evaluates them in one flow launch (`evolutionEq.from_latent`); VMCPDE_FUSED_SAMPLE_EVAL=0 keeps the
    reference's two calls (psi.sample, then evolutionEq(psi, x, t)).  Same samples, same S, F, update to round-off; a psi whose `sample` was
    replaced by the caller is never bypassed."""
    import vmc_pde_amd as V
    dim, N = 6, 2000
    off = np.array([1.0, 0.0, 0.0, 1.0, 0.0, 0.0])
    out = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("VMCPDE_FUSED_SAMPLE_EVAL", fused)
        sampler = V.Sampler(key=3, dim=dim, name="Gauss", mcmc_info={"offset": off, "bound": 0.25})
        vs = V.VarState(sampler, dim, 1, 4, network_args={"intmediate": (5, 5), "offset": off, "latentSpaceName": "Gauss", "dim": dim})
        th = vs.get_parameters().clone()
        th += 0.05 * torch.randn(th.shape, dtype=th.dtype, device=th.device, generator=torch.Generator(device=th.device).manual_seed(2))
        eq = V.EvolutionEquation(dim=dim, name="advection_hamiltonian_wDiss")
        tp = V.TDVP(diagonalShift=1e-4, svdTol=1e-6, solver="cholesky")
        u, info = tp(th, 0.0, vs, eq, nSamplesTDVP=N, nSamplesObs=N, timings=None, integrals=False)
        out[fused] = (u.clone(), tp.S0.clone(), tp.F0.clone(), info["x1"].clone(), info["covar"].clone(), float(info["entropy"]))
    a, b = out["1"], out["0"]
    assert rel_err(a[1], b[1]) < 1e-11 and rel_err(a[2], b[2]) < 1e-10 and rel_err(a[0], b[0]) < 1e-7
    assert rel_err(a[3], b[3]) < 1e-12 and rel_err(a[4], b[4]) < 1e-12 and abs(a[5] - b[5]) < 1e-12
    monkeypatch.setenv("VMCPDE_FUSED_SAMPLE_EVAL", "1")
    assert V.TDVP._fused_sample_eval(vs, eq)
    vs.sample = lambda numSamples: V.VarState.sample(vs, numSamples)
    assert not V.TDVP._fused_sample_eval(vs, eq)
