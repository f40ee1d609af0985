"""End-to-end known answer from the reference's PUBLISHED run (tests/golden/published_diffusion_d8_gauss.npz, extracted from
paper_plot/data_diffusion/dim8_Gauss_.../infos.hdf5): d = 8 Gaussian diffusion, N_TDVP = 10 000, P = 392, Heun with
dt = 1e-7 doubling to 1e-2, 517 steps to t = 5.003.  The whole product stack (Philox sampler -> flow kernels -> RHS ->
MFMA Gram -> spectral solve -> stepper) is run on the same schedule; RNG streams differ from JAX's, so the
comparison is statistical, with the reference's own deviation from the analytic entropy (0.057) as the yardstick."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _run(latent, n_steps, seed=1, dts=None, on_device=True):
    import vmc_pde_amd as V
    dim = 8
    sampler = V.Sampler(key=seed, dim=dim, numChains=30, name=latent, mcmc_info={"offset": np.zeros(dim), "bound": 0.25})
    # 392 = 44 latent + 348 flow parameters <=> 3 blocks x hidden (6,)  
    vs = V.VarState(sampler, dim, seed, 3, network_args={"intmediate": (6,), "offset": np.zeros(dim), "latentSpaceName": latent, "dim": dim})
    stepper = V.FixedStepper(timeStep=5e-8, mode="Heun", maxStep=1e-2, increase_fac=2.0)   # first step 1e-7, doubling (published `times`)
    tdvp = V.TDVP(diagonalizeOnDevice=on_device)
    eq = V.EvolutionEquation(dim=dim, name="diffusion")
    t, out = 0.0, dict(times=[], entropy=[], covar=[], tdvp_error=[], solver_res=[], nu=[])
    for n in range(n_steps):
        if dts is not None:      # replay a recorded (partly adaptive) step-size sequence
            stepper.dt, stepper.increase_fac, stepper.maxStep = float(dts[n]), 1.0, 1.0
        dp, dt, info = stepper.step(0, tdvp, vs.get_parameters(), evolutionEq=eq, psi=vs, nSamplesTDVP=10000, nSamplesObs=10000,
                                    normFunction=lambda v, S: v @ S @ v, timings=None, integrals=None)
        vs.set_parameters(dp)
        out["times"].append(t)
        out["entropy"].append(float(info["entropy"]))
        out["covar"].append(torch.diagonal(info["covar"]).cpu().numpy())
        out["tdvp_error"].append(float(tdvp.tdvp_error))
        out["solver_res"].append(float(tdvp.solverResidual))
        if latent == "Student_t":
            out["nu"].append(float(torch.exp(vs.params["params"]["dist_params"][0]) + 1))
        t += dt
    return vs, {k: np.asarray(v) for k, v in out.items()}


def test_gaussian_diffusion_d8_matches_published_trajectory():
    ref = np.load(os.path.join(GOLD, "published_diffusion_d8_gauss.npz"))
    vs, run = _run("Gauss", len(ref["times"]))
    assert vs.numParameters == ref["ev_first"].shape[0] == 392
    assert np.allclose(run["times"], ref["times"], rtol=0, atol=1e-9)                  # stepper.py:131 schedule
    analytic = 0.5 * 8 * np.log(2 * np.pi * np.e * (1 + 2 * run["times"]))            # visualization.py:188
    dev_ref = np.abs(ref["entropy"] - analytic).max()                                  # 0.057 in the published run
    dev_run = np.abs(run["entropy"] - analytic).max()
    assert dev_ref < 0.06 and dev_run < 2.0 * dev_ref, (dev_ref, dev_run)
    assert np.abs(run["entropy"] - ref["entropy"]).max() < 0.15
    # variance 1 + 2t (visualization.py:162): published diag at t = 5.003 is 10.67..11.19 around 11.006
    assert np.all(np.abs(run["covar"][-1] - (1 + 2 * run["times"][-1])) < 0.6)
    # solver diagnostics in the published range (tdvp_error median 5.1e-5 / max 8.8e-4; residual max 6.5e-11)
    assert np.median(run["tdvp_error"]) < 5e-4 and run["tdvp_error"].max() < 5e-3
    assert np.median(run["solver_res"]) < 1e-9


def test_student_t_diffusion_d8_follows_published_nu():
    # Student-t latent starting at nu = 2 (infinite variance) relaxing towards a Gaussian.  The evolution of the shape
    # parameter nu = exp(dist_params)+1 is nearly deterministic, so it pins the Student-t log-density, its dist_params /
    # L / mu gradients (net.py:23-36) and the chi^2 sampler end to end against the reference's published output.
    # Only the first 40 steps (t <= 0.19) are compared: at nu ~ 2.5 a single heavy-tail outlier can derail any one run
    # (the reference guards this with its NaN exit, tdvp.py:136-141), and the published run changed its step size
    # adaptively between t = 1.6e-3 and 3e-2, which is replayed from the recorded times.  The entropy is not compared:
    # main.py:55-57 notes that the published diffusion runs used another covariance parametrisation (I + A A^T).
    ref = np.load(os.path.join(GOLD, "published_diffusion_d8_student_t.npz"))
    n = 40
    vs, run = _run("Student_t", n, dts=np.diff(ref["times"]), on_device=False)     # host LAPACK eigh as tdvp.py:61-64
    assert vs.numParameters == 393
    assert np.allclose(run["times"], ref["times"][:n], rtol=0, atol=1e-9)
    nu_ref = np.exp(ref["dist_params"][:n]) + 1                                      # recorded after each step (main.py:162,190), as in _run
    assert np.abs(run["nu"] / nu_ref - 1).max() < 0.015, np.abs(run["nu"] / nu_ref - 1).max()
    assert run["nu"][-1] > 2.5                                                       # 2.00 -> 2.55 over these steps in the published run
    # the ansatz cannot follow this flow exactly: the published TDVP error is 0.22 -> 0.11 over these steps
    assert np.all(run["tdvp_error"] < 2 * ref["tdvp_error"][:n] + 0.02) and np.all(run["tdvp_error"] > 0.4 * ref["tdvp_error"][:n])
