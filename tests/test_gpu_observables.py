"""Row a-12 of SURVEY.md section 8: the observables block of TDVP.__call__ (tdvp.py:143-162) by VALUE -- the small-ball
integrals against a closed form and the case nSamplesObs > nSamplesTDVP (tdvp.py:130-134), where the samples and log p are
re-drawn at nSamplesObs while max Eloc still runs over the nSamplesTDVP entries of the solve (tdvp.py:150).
(Injected-point parity of the integrals with the oracle: tests/test_gpu_tdvp.py::test_tdvp_rhs_parity_eigh_path; the
reference's published integral records: tests/test_gpu_published_trajectory.py.)"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gaussian_state(dim, sigma, hidden=(3,), depth=2):
    """Identity flow (output kernels exactly 0) over the latent N(0, sigma^2 I): L = 0, L_diag = log sigma (util.py:21-26)."""
    import vmc_pde_amd as V
    sampler = V.Sampler(key=3, dim=dim, name="Gauss", mcmc_info={"offset": np.zeros(dim), "bound": 0.25})
    vs = V.VarState(sampler, dim, 1, depth, network_args={"intmediate": hidden, "offset": np.zeros(dim), "latentSpaceName": "Gauss", "dim": dim})
    th = vs.get_parameters().clone()
    for path, off, shape in vs.net.layout:
        n = int(np.prod(shape))
        if path[-1] == "L_diag":
            th[off:off + n] = math.log(sigma)
        elif path[-1] == "kernel" and path[-2] == f"Dense_{len(hidden)}":      # output layers of the conditioner MLPs (net.py:58-61)
            th[off:off + n] = 0.0
    vs.set_parameters(th)
    return vs, th


@pytest.mark.parametrize("dim,sigma", [(2, math.sqrt(10.0)), (3, 1.5), (6, 1.0), (8, 1.0)])
def test_small_ball_integrals_known_answer(dim, sigma):
    """p = N(0, sigma^2 I)  ->  integral_{lim}sigma = chi^2_d-CDF(10 lim^2 / sigma^2).  d = 6, sigma^2 = 10 ... the reference's
    steady-state constants (paper_plot_phaseSpaceTempDifference.py:129-133) are this formula at sigma^2 = T = 10, d = 6; d = 8,
    sigma = 1 is the first record of the published diffusion run (0.7363 against 0.7350)."""
    import vmc_pde_amd as V
    from scipy.stats import chi2
    vs, th = _gaussian_state(dim, sigma)
    n_tdvp, n_obs = 2048, 1 << 18
    tp = V.TDVP()
    upd, info = tp(th, 0.0, vs, V.EvolutionEquation(dim=dim, name="diffusion"), nSamplesTDVP=n_tdvp, nSamplesObs=n_obs, timings=None)
    x = torch.randn(1, 4096, dim, dtype=torch.float64, device=vs.device) * sigma
    lp = vs(x)
    exact_lp = -0.5 * (x ** 2).sum(-1) / sigma ** 2 - 0.5 * dim * math.log(2 * math.pi * sigma ** 2)
    assert float((lp - exact_lp).abs().max()) < 1e-12                      # the state really is the Gaussian
    for lim in (1, 0.5, 0.1):
        exact = chi2.cdf(10.0 * lim ** 2 / sigma ** 2, dim)
        # relative standard error of the mean of p over n_obs uniform ball points, from fresh points of the same law
        pts = tp._ball_samples(vs, n_obs, dim)
        pv = torch.exp(vs(lim * math.sqrt(10.0) * pts))[0]
        se = float(pv.std() / pv.mean()) / math.sqrt(n_obs)
        got = float(info[f"integral_{lim}sigma"])
        assert abs(got / exact - 1) < 5 * se + 1e-12, (lim, got, exact, se)
    # entropy of the n_obs re-drawn samples: d/2 log(2 pi e sigma^2) +- sqrt(d/2/n_obs)
    assert abs(float(info["entropy"]) - 0.5 * dim * math.log(2 * math.pi * math.e * sigma ** 2)) < 5 * math.sqrt(0.5 * dim / n_obs)


def test_ball_points_are_uniform_in_the_unit_ball():
    import vmc_pde_amd as V
    vs, th = _gaussian_state(3, 1.0)
    pts = V.TDVP()._ball_samples(vs, 1 << 18, 3)
    assert pts.shape == (1, 1 << 18, 3)
    r = torch.linalg.norm(pts[0], dim=-1)
    assert float(r.max()) <= 1.0
    assert abs(float((r ** 3).mean()) - 0.5) < 5 / math.sqrt(12 * (1 << 18))           # r^d ~ U(0,1)
    assert float(pts[0].mean(0).abs().max()) < 5 * math.sqrt(0.2 / (1 << 18))          # E x_j^2 = 1/(d+2) = 0.2


@pytest.mark.parametrize("n_tdvp,n_obs", [(256, 512), (1000, 4099), (4096, 1 << 16)])
def test_more_observable_samples_than_tdvp_samples(n_tdvp, n_obs):
    """tdvp.py:130-134,143-150 with nSamplesObs > nSamplesTDVP: x1 / covar / entropy / x3..x6 belong to the second, larger sample
    set, max_grad to the Eloc of the TDVP samples (round-2 finding: the kernel read Eloc with the observable sample count)."""
    import vmc_pde_amd as V
    dim = 3
    sampler = V.Sampler(key=5, dim=dim, name="Gauss", mcmc_info={"offset": np.array([0.3, -0.2, 0.1]), "bound": 0.25})
    vs = V.VarState(sampler, dim, 1, 3, network_args={"intmediate": (4,), "offset": np.array([0.3, -0.2, 0.1]), "latentSpaceName": "Gauss", "dim": dim})
    th = vs.get_parameters().clone()
    th += 0.05 * torch.randn(th.shape, dtype=th.dtype, device=th.device, generator=torch.Generator(device=th.device).manual_seed(1))
    drawn, orig = [], vs.sample

    def recording_sample(numSamples):
        x, lp = orig(numSamples=numSamples)
        drawn.append((x.clone(), lp.clone()))
        return x, lp

    vs.sample = recording_sample
    tp = V.TDVP()
    eq = V.EvolutionEquation(dim=dim, name="diffusion_drift")
    for _ in range(2):          # twice: the second call must not see stale state either
        drawn.clear()
        upd, info = tp(th, 0.1, vs, eq, nSamplesTDVP=n_tdvp, nSamplesObs=n_obs, timings=None, integrals=False)
        assert [d[0].shape[1] for d in drawn] == [n_tdvp, n_obs]
        assert not any(k.startswith("integral") for k in info)                 # integrals=False switches them off
        x_t, _ = drawn[0]
        Eloc, _, _ = eq(vs, x_t, 0.1)
        assert Eloc.numel() == n_tdvp and float(info["max_grad"]) == float(Eloc.max())
        x, lp = (a.cpu().numpy()[0] for a in drawn[1])
        m = x.mean(0)
        assert np.allclose(info["x1"].cpu().numpy(), m, rtol=0, atol=1e-13)
        assert np.allclose(info["covar"].cpu().numpy(), (x - m).T @ (x - m) / n_obs, rtol=1e-12, atol=1e-13)
        assert abs(float(info["entropy"]) + lp.mean()) < 1e-13
        for pw in (3, 4, 5, 6):
            ref = ((x - m) ** pw).mean(0)
            assert np.allclose(info[f"x{pw}"].cpu().numpy(), ref, rtol=1e-11, atol=1e-11 * np.abs(ref).max()), pw


def test_observables_abi_takes_eloc_with_its_own_length():
    from vmc_pde_amd import _native as nat
    rng = np.random.default_rng(0)
    n, d = 5000, 2
    dev = torch.device("cuda", 0)
    x, lp = torch.as_tensor(rng.standard_normal((n, d))).to(dev), torch.as_tensor(rng.standard_normal(n)).to(dev)
    ws = torch.zeros(nat.lib().vmcpde_observables_workspace_bytes() // 8, dtype=torch.float64, device=dev)
    out = torch.empty(d * d + 5 * d + 2, dtype=torch.float64, device=dev)
    for n_e in (1, 63, 700, 5000, 20000):
        # the entries behind n_e hold a larger value: reading past the end would show
        el = torch.as_tensor(np.concatenate([rng.standard_normal(n_e), [1e9] * 64])).to(dev)
        nat.call("vmcpde_observables", nat.ptr(x), n, d, nat.ptr(lp), nat.ptr(el), n_e, nat.ptr(out), nat.ptr(ws), ws.numel() * 8, nat.stream())
        assert float(out[-1]) == float(el[:n_e].max()) and abs(float(out[-2]) + float(lp.mean())) < 1e-13
    nat.call("vmcpde_observables", nat.ptr(x), n, d, nat.ptr(lp), None, 0, nat.ptr(out), nat.ptr(ws), ws.numel() * 8, nat.stream())
    assert float(out[-1]) == -math.inf
