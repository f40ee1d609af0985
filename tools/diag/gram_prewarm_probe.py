"""Round-4 verdict item 7b: the Gram runs ~7 % slower in the loop of the reference-default spectral solve (18.4 ms) than in the Cholesky loop
(17.2 ms) -- same kernel, same cycles, lower clock behind the 7 ms of latency-bound launch chains.  Does a short burst of back-to-back fp64
MFMAs right in front of the Gram (a "filler" that brings the matrix pipe's power state up) pay for itself?  Prints the step time and the
Gram's HIP-event bracket for filler lengths 0 .. ~0.7 ms.
    python tools/diag/gram_prewarm_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, bench
import vmc_pde_amd as V
from vmc_pde_amd import _native as nat

dev = torch.device("cuda", 0)
vs, eq, tdvp, theta = bench.build_problem(dev, "C2")
tdvp = V.TDVP()                        # as the reference constructs it: spectral pseudo-inverse
N = bench.WORKLOADS["C2"]["n_samp"]
kw = dict(evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
scratch = torch.empty(512 * 256, dtype=torch.float64, device=dev)
plain = type(tdvp)._gram_launch
for iters in (0, 160, 480, 960, 1920):
    def launch(self, *a, _iters=iters):
        if _iters:
            nat.call("vmcpde_bench_mfma_f64", _iters, 512, nat.ptr(scratch), nat.stream())
        return plain(self, *a)
    type(tdvp)._gram_launch = launch
    ev = {}
    tdvp._event_hook = lambda name, a, b: ev.setdefault(name, []).append((a, b))
    for _ in range(4):
        tdvp(theta, 0.0, **kw)
    torch.cuda.synchronize()
    ev.clear()
    t0 = time.perf_counter()
    for _ in range(12):
        tdvp(theta, 0.0, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 12 * 1e3
    gram = float(np.mean([a.elapsed_time(b) for a, b in ev["gram"]]))
    print(f"filler iters {iters:5d}: step {dt:.3f} ms, gram bracket (filler included) {gram:.3f} ms", flush=True)
type(tdvp)._gram_launch = plain
