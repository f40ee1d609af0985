"""Debug: int8 Gram at C3 inside a Heun step (NaN seen in bench.py --workload C3 --jacobian i8sliced)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import vmc_pde_amd as V
dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
vs, eq, tdvp, theta = bench.build_problem(dev, wl, experimental=True)
tdvp.int8Gram = True
N = bench.WORKLOADS[wl]["n_samp"]
kw = dict(evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
def report(tag, u):
    b = tdvp._lazy["b"]
    JT = tdvp._lazy["JT"]
    P = tdvp._lazy["P"]
    rowmax = JT[:P + 2, :N].abs().amax(1)
    print(tag, "update finite", bool(torch.isfinite(u).all()), "| packed nan", int(torch.isnan(b["packed"]).sum()), "inf", int(torch.isinf(b["packed"]).sum()),
          "| S nan", int(torch.isnan(b["S"]).sum()), "| J finite", bool(torch.isfinite(JT).all()), "row max range", float(rowmax.min()), float(rowmax.max()),
          "| uncentred", tdvp._uncentred_ok, "resid", float(tdvp.solverResidual), flush=True)
for i in range(3):
    u, _ = tdvp(theta, 0.0, **kw)
    report(f"call {i}", u)
y2 = theta + 1e-4 * u
u2, _ = tdvp(y2, 1e-4, **kw)
report("stage-2 state", u2)
tdvp.int8Gram = False
u3, _ = tdvp(y2, 1e-4, **kw)
report("stage-2 state, fp64 gram", u3)
