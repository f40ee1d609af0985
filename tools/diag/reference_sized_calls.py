"""TDVP call times of the reference's own published problem sizes (N = 10 000): d = 8 diffusion (P = 392: eight Hessian directions
per sample) and d = 6 phase space (P = 411: three), per solver, with the split sampling / Eloc+J / solve."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import vmc_pde_amd as V
N = 10000
for label, dim, hidden, eqname, kwnet in (("d=8 diffusion", 8, (4,), "diffusion", {}), ("d=6 phase space", 6, (3,), "advection_hamiltonian_wDiss", {"different_add": True})):
    off = np.zeros(dim)
    sampler = V.Sampler(key=1, dim=dim, name="Gauss", mcmc_info={"offset": off, "bound": 0.25})
    vs = V.VarState(sampler, dim, 1, 4, network_args={"intmediate": hidden, "offset": off, "latentSpaceName": "Gauss", "dim": dim, **kwnet})
    eq = V.EvolutionEquation(dim=dim, name=eqname)
    for name, tdvp in (("cholesky", V.TDVP(diagonalShift=1e-4, svdTol=1e-8, solver="cholesky")), ("eigh (reference default)", V.TDVP())):
        th = vs.get_parameters()
        kw = dict(evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
        for _ in range(5): tdvp(th, 0.0, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): tdvp(th, 0.0, **kw)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        tm = V.util.Timings()
        for _ in range(10): tdvp(th, 0.0, **{**kw, "timings": tm})
        print(f"{label:16s} P={vs.numParameters} {name:24s} {(t1 - t0) / 30 * 1e3:7.3f} ms per call; " +
              ", ".join(f"{k}: {np.median(v) * 1e3:.3f}" for k, v in tm.timing_dict.items()), flush=True)
