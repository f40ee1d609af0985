"""S0 v = J^T (J v) / N of the matrix-free step: two library gemv passes over the whole Jacobian against the same two products taken
sample chunk by sample chunk, so that the second read of a chunk comes from the Infinity Cache."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = torch.device("cuda", 0)
for P, n in [(4213, 65536), (16181, 65536)]:
    JT = torch.randn((P, n), dtype=torch.float64, device=dev)
    v = torch.randn(P, dtype=torch.float64, device=dev)

    def timed(f, reps=10):
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = f()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / reps, out

    t1, z = timed(lambda: v @ JT)
    t2, y = timed(lambda: JT @ z)
    t_full, y0 = timed(lambda: JT @ (v @ JT))
    line = f"P {P} n {n}: v@J {t1:.3f} ms ({P * n * 8 / t1 / 1e9:.2f} TB/s), J@z {t2:.3f} ms ({P * n * 8 / t2 / 1e9:.2f} TB/s), both {t_full:.3f} ms; chunked"
    for ck in (2048, 4096, 8192, 16384):
        def chunked():
            y = torch.zeros(P, dtype=torch.float64, device=dev)
            for c in range(0, n, ck):
                Jc = JT[:, c:c + ck]
                y.addmv_(Jc, v @ Jc)
            return y
        t, yc = timed(chunked)
        line += f" {ck}: {t:.3f} ms"
        err = float((yc - y0).abs().max() / y0.abs().max())
    # one batched call per direction: z as a batch of chunks, y as the sum of a batch
    for ck in (4096,):
        J3 = JT.view(P, n // ck, ck).permute(1, 0, 2)          # chunks x P x ck
        def batched():
            z3 = torch.matmul(v, J3)                            # chunks x ck
            return torch.bmm(J3, z3[:, :, None]).sum(0)[:, 0]
        t, yb = timed(batched)
        line += f"; bmm form {ck}: {t:.3f} ms"
    print(line + f" (diff {err:.1e})", flush=True)
    del JT
