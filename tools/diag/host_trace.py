"""Where the HOST spends a TDVP step of the bench loop (no device synchronisation added): wall time between the Python-level stages."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from vmc_pde_amd import tdvp as T, _native as nat

dev = torch.device("cuda", 0)
vs, eq, tdvp, theta = bench.build_problem(dev, "C2")
kw = dict(evolutionEq=eq, psi=vs, nSamplesTDVP=65536, nSamplesObs=65536, timings=None, observables=False)
marks = []
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); marks.append((label, time.perf_counter() - t)); return r
    setattr(obj, name, g)
wrap(vs, "set_parameters", "set_parameters")
wrap(vs, "sample", "sample")
wrap(tdvp, "get_tdvp_equation", "get_tdvp_equation")
wrap(tdvp, "_gram", "  _gram")
wrap(tdvp, "_check_guard", "  _check_guard")
orig_call = nat.call
def timed_call(name, *a):
    t = time.perf_counter(); r = orig_call(name, *a); marks.append(("    " + name, time.perf_counter() - t)); return r
nat.call = timed_call
T.nat.call = timed_call
for _ in range(3):
    tdvp(theta, 0.0, **kw)
torch.cuda.synchronize()
marks.clear()
ts = []
t00 = time.perf_counter()
for _ in range(10):
    t = time.perf_counter(); tdvp(theta, 0.0, **kw); ts.append(time.perf_counter() - t)
torch.cuda.synchronize()
print("wall per step %.3f ms; host time inside tdvp(): %s ms" % ((time.perf_counter() - t00) / 10 * 1e3, " ".join(f"{x*1e3:.2f}" for x in ts)))
agg = {}
for k, v in marks:
    agg.setdefault(k, []).append(v)
for k, v in agg.items():
    print(f"{k:40s} calls/step {len(v)/10:4.1f}  total/step {sum(v)/10*1e3:8.3f} ms  max {max(v)*1e3:8.3f} ms")
