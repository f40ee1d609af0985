"""Reference-sized problem (d = 6, P = 411, N = 10 000, eigh solver = the reference's semantics) as a user's loop runs it
(observables on): wall time per call and where the host spends it."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import vmc_pde_amd as V
dim, offset, N = 6, np.array([1.0, 0.0, 1.0, 0.0, 1.0, 0.0]), 10000
sampler = V.Sampler(key=1, dim=dim, name="Gauss", mcmc_info={"offset": offset, "bound": 0.25})
vs = V.VarState(sampler, dim, 1, 4, network_args={"intmediate": (3,), "offset": offset, "latentSpaceName": "Gauss", "dim": dim, "different_add": True})
eq = V.EvolutionEquation(dim=dim, name="advection_hamiltonian_wDiss")
for name, tdvp in (("eigh host (default)", V.TDVP()), ("eigh host + SNR", V.TDVP(useSNR=True)), ("cholesky", V.TDVP(diagonalShift=1e-4, svdTol=1e-8, solver="cholesky"))):
    th = vs.get_parameters()
    kw = dict(evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N, timings=None)
    for _ in range(5): tdvp(th, 0.0, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): tdvp(th, 0.0, **kw)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"== {name}: {(t1 - t0) / 30 * 1e3:.3f} ms per TDVP call with observables", flush=True)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20): tdvp(th, 0.0, **kw)
    torch.cuda.synchronize(); pr.disable()
    st = pstats.Stats(pr); st.sort_stats("cumulative")
    import io
    buf = io.StringIO(); pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(12)
    print("\n".join(l for l in buf.getvalue().splitlines() if l.strip() and ("tottime" in l or "{" in l or ".py" in l))[:2600])
