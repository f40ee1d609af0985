"""Long run of the headline configuration (d = 2 Kramers problem, hidden (21,21), P = 4213, 65 536 samples, Cholesky path):
Heun to t = 5 on the reference's step schedule, compared with the exact moments.  Prints a few checkpoints."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import vmc_pde_amd as V
from test_gpu_published_trajectory import _oscillator_moments
shift = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-4
if len(sys.argv) > 2 and sys.argv[2] == "C4":      # BASELINE C4: three damped oscillators, hidden (19,19), P = 4155, 32 768 samples
    dim, offset, N, HID, T_END = 6, np.array([1.0, 0.0, 0.0, 1.0, 0.0, 0.0]), 32768, (19, 19), 2.0
else:
    dim, offset, N, HID, T_END = 2, np.array([1.0, 1.0]), 65536, (21, 21), 5.0
T_END = float(os.environ.get("T_END", T_END))          # (e.g. T_END=1.0 for the dense-engine run of round 6: 147 ms per right-hand side)
sampler = V.Sampler(key=7, dim=dim, name="Gauss", mcmc_info={"offset": offset, "bound": 0.25})
vs = V.VarState(sampler, dim, 1, 4, network_args={"intmediate": HID, "offset": offset, "latentSpaceName": "Gauss", "dim": dim})
eq = V.EvolutionEquation(dim=dim, name="advection_hamiltonian_wDiss")
from vmc_pde_amd.experimental import TDVP as XTDVP       # (= V.TDVP unless INT8_GRAM=1 / MATRIX_FREE=1 switch an experiment on)
if os.environ.get("SOLVER", "cholesky") == "eigh":     # the reference's spectral solve (diagonalShift 0); ENGINE=dense|lowrank|auto picks how (ev, V) are obtained
    tdvp = XTDVP(svdTol=float(os.environ.get("SVD_TOL", "1e-6")), spectralEngine=os.environ.get("ENGINE", "auto"),
                  useSNR=os.environ.get("USE_SNR", "0") == "1",         # USE_SNR=1: the reference's SNR soft cut-off on top (tdvp.py:87-89)
                  matrixFree=os.environ.get("MATRIX_FREE", "0") == "1")
else:
    tdvp = XTDVP(diagonalShift=shift, svdTol=float(os.environ.get("SVD_TOL", "1e-6")), solver="cholesky",
                  int8Gram=os.environ.get("INT8_GRAM", "0") == "1",      # INT8_GRAM=1: the S-build on the int8 matrix pipe (experiment)
                  matrixFree=os.environ.get("MATRIX_FREE", "0") == "1")  # MATRIX_FREE=1: S never assembled (experiment)
engines, ranks = {}, []
modes, bsteps = {}, []
stepper = V.FixedStepper(timeStep=1e-4, mode="Heun", maxStep=1e-2, increase_fac=1.3)
t, n, t0 = 0.0, 0, time.perf_counter()
next_print = 0.5
while t < T_END:
    dp, dt, info = stepper.step(0, tdvp, vs.get_parameters(), evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N,
                                normFunction=lambda v, S: v @ S @ v, timings=None, integrals=None)
    vs.set_parameters(dp)
    t += dt
    n += 1
    if os.environ.get("READ_SNR", "0") == "1" and tdvp.ev is not None:      # as the reference's driver does after every step (main.py:186-187)
        ev_host, snr_host = tdvp.ev.cpu(), tdvp.snr.cpu()
        assert bool(torch.isfinite(snr_host).all())
    if tdvp.choleskyInfo is not None:
        key = f"{tdvp.choleskyInfo.get('engine')}/{tdvp.choleskyInfo.get('mode', '-')}"
        modes[key] = modes.get(key, 0) + 1
        if "block_steps" in tdvp.choleskyInfo:
            bsteps.append(tdvp.choleskyInfo["block_steps"])
    if tdvp.spectralInfo is not None:
        engines[tdvp.spectralInfo["engine"]] = engines.get(tdvp.spectralInfo["engine"], 0) + 1
        if "rank" in tdvp.spectralInfo:
            ranks.append(tdvp.spectralInfo["rank"])
    if t >= next_print or t >= T_END:
        # observables are taken before the step: compare with t - dt; every oscillator (x, p) is the same 2 x 2 problem
        x1, cov = info["x1"].cpu().numpy(), info["covar"].cpu().numpy()
        m_ex = np.concatenate([_oscillator_moments([t - dt], T=10.0, mean0=tuple(offset[2 * o:2 * o + 2]))[0][0] for o in range(dim // 2)])
        c_ex = np.concatenate([np.diagonal(_oscillator_moments([t - dt], T=10.0, mean0=tuple(offset[2 * o:2 * o + 2]))[1][0]) for o in range(dim // 2)])
        print(f"t={t - dt:.2f} step {n}: mean {x1.round(3)} exact {m_ex.round(3)} | cov diag {np.diagonal(cov).round(3)} exact {c_ex.round(3)} "
              f"| entropy {float(info['entropy']):.3f} | {time.perf_counter() - t0:.1f} s", flush=True)
        next_print += 0.5
if engines:
    print(f"spectral engine of the last right-hand side of each step: {engines}; numerical rank {min(ranks) if ranks else '-'} .. {max(ranks) if ranks else '-'}", flush=True)
if modes:
    print(f"regularised solve of the last right-hand side of each step: {modes}; block steps launched {min(bsteps) if bsteps else '-'} .. "
          f"{max(bsteps) if bsteps else '-'} (mean {np.mean(bsteps) if bsteps else 0:.1f}); pivot hints {os.environ.get('VMCPDE_PIVOT_HINTS', '1')}", flush=True)
