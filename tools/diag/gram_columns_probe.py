"""vmcpde_gram_columns (32 columns of S = J^T J / N - m m^T from one pass over the Jacobian): against torch, and its time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vmc_pde_amd import _native as nat
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4213
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda", 0)
Pp, ldj, ldl = (P + 127) // 128 * 128, (N + 15) // 16 * 16, (P + 63) // 64 * 64
g = torch.Generator(device=dev).manual_seed(3)
JT = torch.zeros((Pp, ldj), dtype=torch.float64, device=dev)
JT[:P, :N] = torch.randn((P, N), generator=g, dtype=torch.float64, device=dev) + 0.3
mean = JT[:P, :N].mean(1).contiguous()
rows = torch.randperm(P, generator=g, device=dev)[:32].to(torch.int32).contiguous()
Scol = torch.full((32, ldl), float("nan"), dtype=torch.float64, device=dev)
need = nat.lib().vmcpde_gram_columns_workspace_bytes(P, N)
ws = torch.empty(need // 8, dtype=torch.float64, device=dev)
def run():
    nat.call("vmcpde_gram_columns", nat.ptr(JT), ldj, P, N, nat.ptr(rows), nat.ptr(mean), float(N), nat.ptr(Scol), ldl, None, nat.ptr(ws), need, nat.stream())
run(); torch.cuda.synchronize()
ref = JT[rows.long(), :N] @ JT[:P, :N].T / N - torch.outer(mean[rows.long()], mean)
err = float((Scol[:, :P] - ref).abs().max() / ref.abs().max())
print(f"P {P} N {N}: max rel err {err:.2e}; pad zero: {bool((Scol[:, P:] == 0).all())}")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"  {ms:.3f} ms per panel = {P * N * 8 / ms * 1e-9:.2f} TB/s of Jacobian, {2 * 32 * P * N / ms * 1e-9:.1f} TFLOP/s")
