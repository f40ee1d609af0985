import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device("cuda", 0)
vs, eq, tdvp, theta = bench.build_problem(dev, "C4")
N = 32768
kw = dict(evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
for it in range(4):
    u, _ = tdvp(theta, 0.0, **kw)
    torch.cuda.synchronize()
    print(it, "|u|", float(torch.linalg.norm(u)), "max|u|", float(u.abs().max()), "resid", float(tdvp.solverResidual), "info", int(tdvp.cholInfo.item()), "err", float(tdvp.tdvp_error),
          "Emean", float(tdvp.ElocMean), "Evar", float(tdvp.ElocVar), "ratio", float(tdvp._scal[4]), "floor", float(tdvp._scal[6]), flush=True)
th2 = theta + 1e-4 * u
u2, _ = tdvp(th2, 1e-4, **kw)
torch.cuda.synchronize()
print("stage2 |u|", float(torch.linalg.norm(u2)), "resid", float(tdvp.solverResidual), "info", int(tdvp.cholInfo.item()), "nan in S", bool(torch.isnan(tdvp.S0).any()), "Emean", float(tdvp.ElocMean))
