"""The reference-semantics solve (spectral pseudo-inverse, solver="eigh") at headline size C2: time per TDVP call with the bench's
shifted S (diagonalShift = 1e-4: full rank, dense decomposition on the device / on the host) and with the reference's default
diagonalShift = 0, where S is numerically rank deficient and the rank-revealing engine (csrc/eig.hip) applies."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
vs, eq, tdvp, theta = bench.build_problem(dev, wl)
N = bench.WORKLOADS[wl]["n_samp"]
kw = dict(evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
shift = tdvp.diagonalShift
rows = [("cholesky", "cholesky", None, shift, "auto"), (f"eigh, device, shift {shift:g}", "eigh", None, shift, "auto"),
        ("eigh, shift 0, dense (rocSOLVER)", "eigh", None, 0.0, "dense"), ("eigh, shift 0, rank-revealing", "eigh", None, 0.0, "auto")]
if wl == "C2":
    rows.insert(2, (f"eigh, host LAPACK, shift {shift:g}", "eigh", False, shift, "auto"))
if wl == "C3":
    rows = [r for r in rows if r[4] != "dense" and not r[0].startswith("eigh, device")]       # syevd at P = 16 181 takes many seconds
for label, solver, where, sh, engine in rows:
    tdvp.solver, tdvp.diagonalizeOnDevice, tdvp.diagonalShift, tdvp.spectralEngine = solver, where, sh, engine
    tdvp(theta, 0.0, **kw); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        u, _ = tdvp(theta, 0.0, **kw)
    torch.cuda.synchronize()
    info = tdvp.spectralInfo if solver == "eigh" else None
    print(f"{label:36s} {(time.perf_counter() - t0) / 3 * 1e3:9.2f} ms per TDVP call   residual {float(tdvp.solverResidual):.2e}  tdvp_error {float(tdvp.tdvp_error):.4f}  {info or ''}", flush=True)
