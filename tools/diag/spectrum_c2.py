"""How fast does the spectrum of S decay on the model's own Jacobian?  Counts of eigenvalues above 10^-k of the largest at the bench
workloads (initial parameters and after a few Heun steps), and rocSOLVER's time for the full decomposition."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from vmc_pde_amd import FixedStepper
dev = torch.device("cuda", 0)
for wl in sys.argv[1:] or ["C2"]:
    vs, eq, tdvp, theta = bench.build_problem(dev, wl)
    N = bench.WORKLOADS[wl]["n_samp"]
    kw = dict(evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N, timings=None)
    stepper = FixedStepper(timeStep=1e-5, mode="Heun", maxStep=1e-2, increase_fac=1.3)
    t = 0.0
    for stage in range(3):
        tdvp(theta, t, observables=False, **kw)
        S = tdvp.S0.clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ev = torch.linalg.eigvalsh(S)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        ev2, V = torch.linalg.eigh(S)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        lmax = float(ev[-1])
        counts = {k: int((ev > lmax * 10.0 ** -k).sum()) for k in (4, 6, 8, 10, 12, 13, 14, 15, 16)}
        print(f"{wl} t={t:.3g}: P={S.shape[0]} lmax={lmax:.3e} min={float(ev[0]):.2e}; #ev > 1e-k lmax: {counts}; eigvalsh {1e3*(t1-t0):.1f} ms, eigh {1e3*(t2-t1):.1f} ms", flush=True)
        for _ in range(20):
            theta, dt, _ = stepper.step(t, tdvp, theta, **kw)
            vs.set_parameters(theta); t += dt
