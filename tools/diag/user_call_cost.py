"""What a call of a user's loop costs beyond the benchmark's step (C2): observables and small-ball integrals on / off, synchronous as the
reference's loop is (it reads the info record every step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device("cuda", 0)
vs, eq, tdvp, theta = bench.build_problem(dev, "C2")
N = bench.WORKLOADS["C2"]["n_samp"]
for label, kw in (("observables off (the benchmark's step)", dict(observables=False)),
                  ("observables on, integrals off", dict(integrals=False)),
                  ("observables and integrals on (the reference's loop)", dict())):
    for _ in range(6):
        out = tdvp(theta, 0.0, evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N, timings=None, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        out = tdvp(theta, 0.0, evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N, timings=None, **kw)
        if isinstance(out[1], dict) and "entropy" in out[1]:
            float(out[1]["entropy"])                      # (the loop reads its record)
    torch.cuda.synchronize()
    print(f"{label}: {1e3 * (time.perf_counter() - t0) / 20:.2f} ms per call", flush=True)
