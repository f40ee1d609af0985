"""The int8 sliced Gram on the MODEL's Jacobian (bench workloads) against the fp64 kernel on the same samples: rel-Frobenius of S0 and F0,
the update in the S-norm, and the heavy-tail figure the engine watches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
for wl in sys.argv[1:] or ["C2", "C4"]:
    out = {}
    for flag in (False, True):
        vs, eq, tdvp, theta = bench.build_problem(torch.device("cuda", 0), wl, experimental=True)
        tdvp.int8Gram = flag
        N = bench.WORKLOADS[wl]["n_samp"]
        kw = dict(evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
        for _ in range(3):                       # third call: steady state (lagged means, uncentred Gram with the appended rows)
            u, _ = tdvp(theta, 0.0, **kw)
        tdvp.check()
        out[flag] = (tdvp.S0.clone(), tdvp.F0.clone(), u.clone(), float(tdvp._i8_host[0]) if flag else None, tdvp._int8_still_ok() if flag else None)
    (S, F, u, _, _), (S8, F8, u8, ratio, ok) = out[False], out[True]
    d = torch.sqrt(torch.diag(S)).clamp_min(1e-300)
    print(f"{wl}: rel-Frobenius S {float(torch.linalg.norm(S8 - S) / torch.linalg.norm(S)):.2e}, correlation form "
          f"{float(torch.linalg.norm((S8 - S) / torch.outer(d, d)) / torch.linalg.norm(S / torch.outer(d, d))):.2e}, F {float(torch.linalg.norm(F8 - F) / torch.linalg.norm(F)):.2e}, "
          f"update (S-norm) {float(torch.linalg.norm(S @ (u8 - u)) / torch.linalg.norm(F)):.2e}; largest row max / rms {ratio:.3g} (limit {tdvp._int8_tail_limit():.3g}, int8 kept: {ok})", flush=True)
