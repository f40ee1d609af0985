"""Pins the CPU oracle (oracle/vmc_oracle.py) against everything the reference itself fixes: the analytic known
answers KA-1..KA-4 and the parameter counts KA-7 of SURVEY.md section 4.  (Vector-level parity with the JAX code is
unpinned: the reference cannot run here and ships no fixtures.)"""
import math

import numpy as np
import pytest
import torch

from oracle import vmc_oracle as O
from helpers import make_spec, random_theta


def test_ka7_parameter_counts():
    # main.py:38,73 'mwe': d=2, 4 blocks, hidden (1,) -> 37 ; d=6, 4 blocks, (3,), different_add -> 411 (published ev width)
    assert O.num_parameters(make_spec(2, 4, (1,))) == 37
    assert O.num_parameters(make_spec(6, 4, (3,), coupling="different_add")) == 411
    # SURVEY 8a-3 bench nets
    assert O.num_parameters(make_spec(2, 4, (21, 21))) == 4213
    assert O.num_parameters(make_spec(2, 4, (170,))) == 4093
    assert O.num_parameters(make_spec(6, 4, (19, 19))) == 4155
    assert O.num_parameters(make_spec(8, 3, (6,))) == 392          # the published d=8 Gauss run
    assert O.num_parameters(make_spec(8, 3, (6,), latent="Student_t")) == 393


def test_layout_is_sorted_pytree_order():
    spec = make_spec(4, 11, (2,))
    names = [n for n, _ in O.param_layout(spec)]
    assert names[:4] == ["L", "L_diag", "dist_params", "mu"]
    blocks = [n.split("/")[1] for n in names[4:]]
    order = list(dict.fromkeys(blocks))
    assert order == sorted(order) and order.index("blocks_10") < order.index("blocks_2")   # string sort
    first = [n for n in names if n.startswith("myINN/blocks_0/")]
    assert first[0].endswith("s1/Dense_0/bias") and first[1].endswith("s1/Dense_0/kernel")


@pytest.mark.parametrize("dim", [2, 3, 6])
def test_ka1_identity_flow_limit(dim):
    # all output kernels zero -> s == 0 -> logp = standard normal about offset (net.py:96,107,212-213)
    off = np.linspace(-1, 1, dim)
    spec = make_spec(dim, 3, (3,), offset=off)
    th = O.init_parameters(spec, 5, out_scale=0.0)
    vs = O.VarStateOracle(spec, th)
    x = torch.randn(1, 9, dim, dtype=torch.float64)
    lp, g, J = vs(x, "eval_coordgrads")
    r = x - torch.as_tensor(off)
    assert torch.allclose(lp, -0.5 * (r ** 2).sum(-1) - 0.5 * dim * math.log(2 * math.pi), atol=1e-13)
    assert torch.allclose(g, -r, atol=1e-13)
    assert torch.allclose(vs.hessian(x), -torch.eye(dim, dtype=torch.float64).expand(1, 9, dim, dim), atol=1e-13)
    eq = O.EvolutionEquationOracle(dim, "diffusion")
    E, _, _ = eq(vs, x, 0.0)
    assert torch.allclose(E, (r ** 2).sum(-1) - dim, atol=1e-12)     # D(|x-offset|^2 - d), D = 1


@pytest.mark.parametrize("coupling", ["no_add", "different_add", "jac_eq_1", "plain"])
@pytest.mark.parametrize("latent", ["Gauss", "Student_t"])
def test_ka2_invertibility(coupling, latent):
    spec = make_spec(4, 3, (3,), latent=latent, coupling=coupling, offset=[0.3, -0.2, 0.1, 0.0])
    th = random_theta(spec, 3)
    vs = O.VarStateOracle(spec, th)
    z = torch.randn(1, 11, 4, dtype=torch.float64)
    x, lp = vs.flow_sample(z)                                   # inverse flow, net.py:214-217
    zb, lp_f = vmap_fwd(spec, th, x[0])
    assert torch.allclose(zb, z[0], atol=1e-11)                 # inv(fwd(x)) == x (main.py:77-96)
    assert torch.allclose(vs(x, "eval")[0], lp[0], atol=1e-10)  # logjac_fwd + logjac_inv = 0


def vmap_fwd(spec, th, x):
    ys, lps = [], []
    for xi in x:
        y, lp = O.inn_w_prob(spec, th, xi, evaluate=False, inv=False)
        ys.append(y)
        lps.append(lp)
    return torch.stack(ys), torch.stack(lps)


def test_ka3_normalisation_on_grid():
    # main.py:100-105: 200^2 grid, integral of exp(logp) ~ 1
    spec = make_spec(2, 2, (2,))
    th = random_theta(spec, 2, out_scale=0.004, latent_scale=0.2)
    vs = O.VarStateOracle(spec, th)
    b, n = 10.0, 200
    w = 2 * b / n
    v = np.arange(-b, b, w)
    X, Y = np.meshgrid(v, v)
    coords = torch.as_tensor(np.stack([X, Y], -1).reshape(1, -1, 2))
    integral = float((w * w * torch.exp(vs(coords, "eval"))).sum())
    assert abs(integral - 1.0) < 1e-3


def test_ka4_gaussian_diffusion_is_exact_in_L_diag():
    # restricted to the L_diag block at the identity flow: S = 2 I, F = 2 D 1, theta_dot = D 1, tdvp_error -> 0  (N -> inf)
    dim, N = 3, 60000
    spec = make_spec(dim, 2, (2,))
    th = O.init_parameters(spec, 1, out_scale=0.0)
    vs = O.VarStateOracle(spec, th)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, N, dim, generator=g, dtype=torch.float64)
    E, J, lp = O.EvolutionEquationOracle(dim, "diffusion")(vs, x, 0.0)
    names = [n for n, s in O.param_layout(spec) for _ in range(int(np.prod(s)))]
    idx = [i for i, n in enumerate(names) if n == "L_diag"]
    Jd = J[:, :, idx]
    t = O.TDVPOracle()
    upd, resid, err = t.solve(E, Jd, lp)
    assert torch.allclose(t.S0, 2 * torch.eye(dim, dtype=torch.float64), atol=0.06)
    assert torch.allclose(t.F0, 2 * torch.ones(dim, dtype=torch.float64), atol=0.06)
    assert torch.allclose(upd, torch.ones(dim, dtype=torch.float64), atol=0.03)
    assert abs(float(err)) < 5e-3 and float(resid) < 1e-10


def test_student_t_has_no_logdet_term():
    # net.py:35-36: the reference's Student-t omits log det S
    spec = make_spec(2, 1, (2,), latent="Student_t")
    th = O.init_parameters(spec, 1, out_scale=0.0)
    p = O.unflatten(spec, th)
    a = O.latent_logpdf(spec, p, torch.zeros(2, dtype=torch.float64))
    th2 = th.clone()
    th2[1:3] = 0.7   # L_diag
    b = O.latent_logpdf(spec, O.unflatten(spec, th2), torch.zeros(2, dtype=torch.float64))
    assert torch.allclose(a, b)


def test_mcmc_restatement_stationary_density():
    # sampler.py:20-22: stationary density 0.5(1+cos(pi r)), r = min(1, 4|x-offset|): mean radius is analytic
    rng = np.random.default_rng(3)
    chains, steps = 64, 4000
    off = np.array([0.25, 0.25])
    st0 = O.radial_update_prop(rng.random(chains), rng.random(chains), 0.25, off)
    out, nacc = O.mcmc_chain(st0, rng.random((steps, chains)), rng.random((steps, chains)), rng.random((steps, chains)), 0.25, off)
    r = 4 * np.linalg.norm(out - off, axis=1)
    # E[r] under density ~ (1+cos(pi r)) r dr on [0,1]
    rr = np.linspace(0, 1, 20001)
    w = (1 + np.cos(np.pi * rr)) * rr
    expect = np.trapezoid(w * rr, rr) / np.trapezoid(w, rr)
    assert abs(r.mean() - expect) < 0.01
    assert 0 < nacc < steps * chains


def test_distribute_sampling_matches_reference_formula():
    for n, size in [(10000, 1), (10001, 4), (65536, 8), (7, 3)]:
        per = [O.distribute_sampling(n, size, r)[0] for r in range(size)]
        assert sum(per) == n and max(per) - min(per) <= 1
    # chains variant (mpi_wrapper.py:94-110)
    spc, glob = O.distribute_sampling(10000, 1, 0, localDevices=1, numChainsPerDevice=30)
    assert spc == 334 and glob == 334 * 30


@pytest.mark.parametrize("dim,sigma", [(2, math.sqrt(10.0)), (3, 1.5), (6, 1.0)])
def test_small_ball_integrals_known_answer(dim, sigma):
    """tdvp.py:152-162 on a density whose ball probabilities are known: identity flow (output kernels 0), latent covariance
    sigma^2 I (L = 0, L_diag = log sigma, util.py:21-26), mu = offset = 0  ->  p = N(0, sigma^2 I) and
    integral_{lim}sigma = P(|x| < lim sqrt(10)) = chi^2_d-CDF(10 lim^2 / sigma^2).  d = 6, sigma = 1 at lim = 1 is the first
    entry of the steady-state constants the reference's plot script holds (paper_plot_phaseSpaceTempDifference.py:129-133:
    0.0143877 / 2.96478e-4 / 2.07554e-8 are the chi^2_6-CDF at 1, 0.25, 0.01 -- the T = 10 steady state has variance 10)."""
    from scipy.stats import chi2
    spec = make_spec(dim, 2, (3,))
    th = O.init_parameters(spec, 5, out_scale=0.0)
    p = O.unflatten(spec, th)
    p["L_diag"] = torch.full_like(p["L_diag"], math.log(sigma))
    th = O.flatten(spec, p)
    vs = O.VarStateOracle(spec, th)
    g = torch.Generator().manual_seed(dim)
    n = 40000
    ball = O.TDVPOracle.ball_points(torch.randn(1, n, dim, generator=g, dtype=torch.float64), torch.rand(1, n, generator=g, dtype=torch.float64))
    r = torch.linalg.norm(ball[0], dim=-1)
    assert float(r.max()) <= 1.0 and abs(float((r ** dim).mean()) - 0.5) < 0.01          # uniform in the unit ball: r^d ~ U(0,1)
    ints = O.TDVPOracle.small_ball_integrals(vs, ball)
    for lim in (1, 0.5, 0.1):
        exact = chi2.cdf(10.0 * lim ** 2 / sigma ** 2, dim)
        # Monte-Carlo error of the mean of p over the ball, from the sample itself
        R = lim * math.sqrt(10.0)
        pv = torch.exp(vs(R * ball))[0]
        se = float(pv.std() / pv.mean()) / math.sqrt(n)
        assert abs(float(ints[f"integral_{lim}sigma"]) / exact - 1) < 5 * se + 1e-12, (lim, se)
    assert abs(chi2.cdf(1.0, 6) - 0.0143877) < 1e-7 and abs(chi2.cdf(0.25, 6) - 2.96478e-4) < 1e-9 and abs(chi2.cdf(0.01, 6) - 2.07554e-8) < 1e-13
