"""Timing / sanity probe of the fp32 Gram variants (jt_f32 = 1 widening, 2 fp32 MFMA blocks) on random data."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmc_pde_amd import _native as nat
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4213
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda", 0)
Pp, ldj = (P + 127) // 128 * 128, (N + 15) // 16 * 16
g = torch.Generator(device=dev).manual_seed(7)
JT = torch.zeros((Pp, ldj), dtype=torch.float32, device=dev)
JT[:P, :N] = torch.randn((P, N), generator=g, dtype=torch.float32, device=dev)
npk = nat.lib().vmcpde_gram_packed_doubles(P)
need = nat.lib().vmcpde_gram_workspace_bytes(P, N)
ws = torch.empty(max(need // 8, 1), dtype=torch.float64, device=dev)
res = {}
for mode in (1, 2):
    packed = torch.full((npk,), float("nan"), dtype=torch.float64, device=dev)
    ts = []
    for it in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        nat.call("vmcpde_gram", nat.ptr(JT), ldj, mode, P, N, None, None, nat.ptr(packed), nat.ptr(ws), ws.numel() * 8, nat.stream())
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    res[mode] = packed.view(-1, 128, 128)
    bad = ~torch.isfinite(res[mode])
    print(f"mode {mode}: {min(ts[1:]):.3f} ms  {N*P*(P+1)/min(ts[1:])*1e-9:.1f} TFLOP/s  non-finite {int(bad.sum())}"
          f" in tiles {torch.nonzero(bad.flatten(1).any(1)).flatten()[:8].tolist()}", flush=True)
low = torch.tril(torch.ones(128, 128, dtype=torch.bool, device=dev))
d = (res[2] - res[1])
print("rel diff (lower triangles)", float(torch.linalg.norm(d[:, low]) / torch.linalg.norm(res[1][:, low])))
