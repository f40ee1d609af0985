"""How many modes the regularised solve (S0 + D) u = F has to resolve: eigenvalue counts of D^-1/2 S0 D^-1/2 (D = diagonalShift
diag(S0) + floor, the matrix whose pivoted low-rank factor the Woodbury route preconditions with) above a range of thresholds, next
to the counts for S0 / lambda_max, at the bench workloads.  Decides whether a relaxed pivoting tolerance would bring a full-rank S
(C4) within reach of the low-rank route: the refinement contracts by the largest scaled eigenvalue left out of the factor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device("cuda", 0)
for wl in sys.argv[1:] or ["C2", "C4"]:
    vs, eq, tdvp, theta = bench.build_problem(dev, wl)
    N = bench.WORKLOADS[wl]["n_samp"]
    shift = float(tdvp.diagonalShift)
    tdvp(theta, 0.0, evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
    S = tdvp.S0.clone(); P = S.shape[0]
    d = S.diagonal().clone()
    D = shift * d + (tdvp._floor if tdvp._floor is not None else 0.0)
    ev = torch.linalg.eigvalsh(S).flip(0)
    sc = torch.linalg.eigvalsh(S / torch.sqrt(D)[:, None] / torch.sqrt(D)[None, :]).flip(0)
    ths = [1e2, 1.0, 1e-1, 1e-2, 1e-3, 1e-4, 1e-6]
    print(f"{wl}: P {P}, shift {shift:g}, diag S in [{float(d.min()):.2e}, {float(d.max()):.2e}], lambda_max {float(ev[0]):.3e}", flush=True)
    print("   S0 / lambda_max above " + "  ".join(f"{t:g}: {int((ev > t * ev[0]).sum())}" for t in [1e-4, 1e-6, 1e-8, 1e-10, 1e-12, 1e-14, 1e-16]))
    print("   D^-1/2 S0 D^-1/2 (largest %.3e) above " % float(sc[0]) + "  ".join(f"{t:g}: {int((sc > t).sum())}" for t in ths), flush=True)
