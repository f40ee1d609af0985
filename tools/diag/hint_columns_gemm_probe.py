"""All columns S[:, H] of the hinted pivots in ONE pass over the Jacobian (a gather of |H| rows and a library GEMM, batched over sample
chunks) against one colpanel pass per 32 columns: time of the gather, the GEMM variants and the reduction, at the bench workloads."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = torch.device("cuda", 0)
for P, n, rH in [(4213, 65536, 224), (4213, 65536, 256), (16181, 65536, 288)]:
    ldj = n
    JT = torch.randn((P, ldj), dtype=torch.float64, device=dev)
    H = torch.randperm(P, device=dev)[:rH]

    def timed(f, reps=5):
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = f()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / reps, out

    t_g, JH = timed(lambda: JT[H].contiguous())
    t_mm, S1 = timed(lambda: JH @ JT.T)
    res = {}
    for ck in (1024, 2048, 4096, 8192):
        A3 = JH.view(rH, n // ck, ck).permute(1, 0, 2)
        B3 = JT.view(P, n // ck, ck).permute(1, 2, 0)
        t_b, S2 = timed(lambda: torch.bmm(A3, B3).sum(0))
        res[ck] = t_b
        err = float((S2 - S1).abs().max() / S1.abs().max())
    fl = 2.0 * rH * P * n
    print(f"P {P} n {n} rH {rH}: gather {t_g:.3f} ms, one mm {t_mm:.2f} ms ({fl / t_mm / 1e9:.1f} TF), bmm+sum by chunk " +
          ", ".join(f"{ck}: {t:.2f} ms ({fl / t / 1e9:.1f} TF)" for ck, t in res.items()) + f", diff {err:.1e}", flush=True)
    del JT, JH, S1, S2, A3, B3
