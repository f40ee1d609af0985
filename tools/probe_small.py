"""Step time of a reference-sized problem (d = 6, P = 411, N = 10 000): where a small run spends its time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vmc_pde_amd as V
dim, offset, N = 6, np.array([1.0, 0.0, 1.0, 0.0, 1.0, 0.0]), 10000
sampler = V.Sampler(key=1, dim=dim, name="Gauss", mcmc_info={"offset": offset, "bound": 0.25})
vs = V.VarState(sampler, dim, 1, 4, network_args={"intmediate": (3,), "offset": offset, "latentSpaceName": "Gauss", "dim": dim, "different_add": True})
eq = V.EvolutionEquation(dim=dim, name="advection_hamiltonian_wDiss")
for name, tdvp in (("cholesky", V.TDVP(diagonalShift=1e-4, svdTol=1e-8, solver="cholesky")), ("eigh host", V.TDVP(spectralEngine="dense", diagonalizeOnDevice=False)), ("eigh device", V.TDVP(spectralEngine="dense", diagonalizeOnDevice=True)),
                   ("eigh default", V.TDVP()), ("eigh rank-revealing engine forced", V.TDVP(spectralEngine="lowrank"))):
    th = vs.get_parameters()
    kw = dict(evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
    for _ in range(5): tdvp(th, 0.0, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): tdvp(th, 0.0, **kw)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    tm = V.util.Timings()
    for _ in range(10): tdvp(th, 0.0, **{**kw, "timings": tm})
    print(f"{name} [{tdvp.spectralInfo}]: {(t1 - t0) / 50 * 1e3:.3f} ms per TDVP call (P={vs.numParameters}, N={N}); sections [ms] " +
          ", ".join(f"{k}: {np.median(v) * 1e3:.3f}" for k, v in tm.timing_dict.items()), flush=True)
