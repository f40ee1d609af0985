"""Long run of the headline configuration (d = 2 Kramers problem, hidden (21,21), P = 4213, 65 536 samples, Cholesky path):
Heun to t = 5 on the reference's step schedule, compared with the exact moments.  Prints a few checkpoints."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import vmc_pde_amd as V
from test_gpu_published_trajectory import _oscillator_moments
dim, offset, N = 2, np.array([1.0, 1.0]), 65536
shift = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-4
sampler = V.Sampler(key=7, dim=dim, name="Gauss", mcmc_info={"offset": offset, "bound": 0.25})
vs = V.VarState(sampler, dim, 1, 4, network_args={"intmediate": (21, 21), "offset": offset, "latentSpaceName": "Gauss", "dim": dim})
eq = V.EvolutionEquation(dim=dim, name="advection_hamiltonian_wDiss")
tdvp = V.TDVP(diagonalShift=shift, svdTol=float(os.environ.get("SVD_TOL", "1e-6")))
stepper = V.FixedStepper(timeStep=1e-4, mode="Heun", maxStep=1e-2, increase_fac=1.3)
t, n, t0 = 0.0, 0, time.perf_counter()
next_print = 0.5
while t < 5.0:
    dp, dt, info = stepper.step(0, tdvp, vs.get_parameters(), evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N,
                                normFunction=lambda v, S: v @ S @ v, timings=None, integrals=None)
    vs.set_parameters(dp)
    t += dt
    n += 1
    if t >= next_print or t >= 5.0:
        m_ex, c_ex = _oscillator_moments([t], T=10.0, mean0=(1.0, 1.0))
        # observables are taken before the step: compare with t - dt
        m_ex, c_ex = _oscillator_moments([t - dt], T=10.0, mean0=(1.0, 1.0))
        x1, cov = info["x1"].cpu().numpy(), info["covar"].cpu().numpy()
        print(f"t={t - dt:.2f} step {n}: mean {x1.round(3)} exact {m_ex[0].round(3)} | cov diag {np.diagonal(cov).round(3)} exact {np.diagonal(c_ex[0]).round(3)} "
              f"| cov01 {cov[0,1]:.3f} exact {c_ex[0][0,1]:.3f} | entropy {float(info['entropy']):.3f} | {time.perf_counter() - t0:.1f} s", flush=True)
        next_print += 0.5
