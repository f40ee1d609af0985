"""Round 5: what `tdvpEq.snr` costs after a reference-default TDVP() call at the headline size (the reference's driver reads it after every
step, main.py:187): the r x P x N product over the Ritz vectors against the weighted Gram + P^3 product of the dense formula."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import vmc_pde_amd as V
dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
vs, eq, _, theta = bench.build_problem(dev, wl)
N = bench.WORKLOADS[wl]["n_samp"]
tp = V.TDVP()
kw = dict(evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
for mode in ("1", "0", "1"):
    os.environ["VMCPDE_RHOVAR_LOWRANK"] = mode
    ts, tc = [], []
    for it in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tp(theta, 0.0, **kw)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        s = tp.snr
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if it >= 2:
            tc.append((t1 - t0) * 1e3); ts.append((t2 - t1) * 1e3)
    print(f"{wl} VMCPDE_RHOVAR_LOWRANK={mode}: call {sorted(tc)[len(tc)//2]:.2f} ms, snr {sorted(ts)[len(ts)//2]:.2f} ms (engine {tp.spectralInfo['engine']}, rank {tp.spectralInfo.get('rank')}, "
          f"snr finite {bool(torch.isfinite(s).all())}, nonzero {int((s != 0).sum())})", flush=True)
