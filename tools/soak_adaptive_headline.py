"""Soak of the low-rank regularised solve (static plan, pivot hints) under an irregular call sequence: AdaptiveHeun at the headline size
(d = 2 Kramers problem, hidden (21,21), P = 4213, 65 536 samples) -- five right-hand sides per attempt, rejected attempts, step sizes
that jump by up to 2x -- against the same run with the full factorisation.  Same seeds, so both runs draw the same samples as long as
they take the same decisions; prints the step-size history, the solve modes and the distance between the two parameter trajectories."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vmc_pde_amd as V
NSTEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
TOL = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-5
N, HID, dim, offset = 65536, (21, 21), 2, np.array([1.0, 1.0])


def run(engine, matrix_free=False):
    sampler = V.Sampler(key=7, dim=dim, name="Gauss", mcmc_info={"offset": offset, "bound": 0.25})
    vs = V.VarState(sampler, dim, 1, 4, network_args={"intmediate": HID, "offset": offset, "latentSpaceName": "Gauss", "dim": dim})
    eq = V.EvolutionEquation(dim=dim, name="advection_hamiltonian_wDiss")
    from vmc_pde_amd.experimental import TDVP as XTDVP
    tdvp = XTDVP(diagonalShift=1e-4, svdTol=1e-6, solver="cholesky", choleskyEngine=engine, matrixFree=matrix_free)
    stepper = V.AdaptiveHeun(timeStep=1e-4, tol=TOL, maxStep=1e-2)
    t, dts, modes, thetas = 0.0, [], {}, []
    calls = [0]
    inner = tdvp.__call__

    class Counting:                       # (counts right-hand sides: 5 per attempt)
        def __getattr__(self, k):
            return getattr(tdvp, k)

        def __call__(self, *a, **kw):
            calls[0] += 1
            return inner(*a, **kw)
    f = Counting()
    t0 = time.perf_counter()
    for n in range(NSTEPS):
        dp, dt, info = stepper.step(t, f, vs.get_parameters(), evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N,
                                    normFunction=lambda v, S: v @ S @ v, timings=None, integrals=None)
        vs.set_parameters(dp)
        t += dt
        dts.append(dt)
        key = f"{(tdvp.choleskyInfo or {}).get('engine')}/{(tdvp.choleskyInfo or {}).get('mode', '-')}"
        modes[key] = modes.get(key, 0) + 1
        thetas.append(dp.clone())
    torch.cuda.synchronize()
    print(f"{engine}{' matrix-free' if matrix_free else ''}: {NSTEPS} accepted steps to t = {t:.5f} in {time.perf_counter() - t0:.1f} s, "
          f"{calls[0]} right-hand sides ({calls[0] // 5 - NSTEPS} rejected attempts); "
          f"dt {dts[0]:.2e} .. {max(dts):.2e} (last {dts[-1]:.2e}); last solve of each step {modes}; "
          f"mean {info['x1'].cpu().numpy().round(4)} entropy {float(info['entropy']):.4f}", flush=True)
    return t, dts, thetas


ref = run("dense")
for eng, mf in (("auto", False), ("auto", True)):
    got = run(eng, mf)
    same = [abs(a - b) <= 1e-12 * abs(b) for a, b in zip(got[1], ref[1])]
    k = same.index(False) if False in same else len(same)
    d = [float(torch.linalg.norm(a - b) / torch.linalg.norm(b - ref[2][0] + 1e-300)) for a, b in zip(got[2][1:], ref[2][1:])]
    print(f"   step sizes identical to the dense run for the first {k} of {len(same)} steps; |theta - theta_dense| / |theta_dense - theta_0| after "
          f"steps 2, {len(d) // 2 + 1}, {len(d) + 1}: {d[0]:.2e}, {d[len(d) // 2]:.2e}, {d[-1]:.2e}", flush=True)
