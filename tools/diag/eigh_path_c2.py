"""The reference-semantics solve (spectral pseudo-inverse, solver="eigh") at headline size C2: time per TDVP call with the
eigendecomposition on the device (default from 640 parameters on) and on the host (the reference's default)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device("cuda", 0)
vs, eq, tdvp, theta = bench.build_problem(dev, "C2")
kw = dict(evolutionEq=eq, psi=vs, nSamplesTDVP=65536, nSamplesObs=65536, timings=None, observables=False)
for label, solver, where in (("cholesky", "cholesky", None), ("eigh, device (default)", "eigh", None), ("eigh, host LAPACK", "eigh", False)):
    tdvp.solver, tdvp.diagonalizeOnDevice = solver, where
    tdvp(theta, 0.0, **kw); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        u, _ = tdvp(theta, 0.0, **kw)
    torch.cuda.synchronize()
    print(f"{label:28s} {(time.perf_counter() - t0) / 3 * 1e3:9.2f} ms per TDVP call   residual {float(tdvp.solverResidual):.2e}  tdvp_error {float(tdvp.tdvp_error):.4f}", flush=True)
