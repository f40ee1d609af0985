"""Does the asynchronous health record (pinned D2H copy + event at the end of a step, host wait at the start of the next Gram)
cost GPU time at the step boundary?  Times the bench loop with and without it, and the host's wait."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from vmc_pde_amd import tdvp as T

dev = torch.device("cuda", 0)
vs, eq, tdvp, theta = bench.build_problem(dev, "C2")
kw = dict(evolutionEq=eq, psi=vs, nSamplesTDVP=65536, nSamplesObs=65536, timings=None, observables=False)


def run(label, steps=40):
    for _ in range(3):
        tdvp(theta, 0.0, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tdvp(theta, 0.0, **kw)
    torch.cuda.synchronize()
    print(f"{label:40s} {(time.perf_counter() - t0) / steps * 1e3:8.3f} ms/step", flush=True)


run("default")
waits = []
orig_check = T.TDVP._check_guard
def timed_check(self, when):
    t = time.perf_counter(); orig_check(self, when); waits.append(time.perf_counter() - t)
T.TDVP._check_guard = timed_check
run("default (timing the host wait)")
print("host wait in _check_guard: median %.2f ms" % (sorted(waits)[len(waits) // 2] * 1e3))
T.TDVP._check_guard = orig_check
orig_rec = T.TDVP._record_guard
T.TDVP._record_guard = lambda self, out: None
run("no health record")
tdvp.gramShift = "uncentred"
run("no health record, no ratio check")
T.TDVP._record_guard = orig_rec
run("health record, no ratio check")
tdvp.gramShift = "auto"
run("default again")
