"""Timing probe: the fp64 MFMA Gram kernel on a random parameter-major Jacobian (default C2 shape 4213 x 65536)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmc_pde_amd import _native as nat
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4213
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
centre = (sys.argv[4] != "nocentre") if len(sys.argv) > 4 else True
dev = torch.device("cuda", 0)
Pp, ldj = (P + 127) // 128 * 128, (N + 15) // 16 * 16
g = torch.Generator(device=dev).manual_seed(7)
JT = torch.zeros((Pp, ldj), dtype=torch.float64, device=dev)
JT[:P, :N] = torch.randn((P, N), generator=g, dtype=torch.float64, device=dev)
c = JT[:P, :N].mean(1).contiguous()
packed = torch.empty(nat.lib().vmcpde_gram_packed_doubles(P), dtype=torch.float64, device=dev)
need = nat.lib().vmcpde_gram_workspace_bytes(P, N)
ws = torch.empty(max(need // 8, 1), dtype=torch.float64, device=dev)
ts = []
for it in range(reps + 1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    nat.call("vmcpde_gram", nat.ptr(JT), ldj, 0, P, N, nat.ptr(c) if centre else None, None, nat.ptr(packed), nat.ptr(ws), ws.numel() * 8, nat.stream())
    e1.record()
    torch.cuda.synchronize()
    if it:
        ts.append(e0.elapsed_time(e1))
ms = sorted(ts)[len(ts) // 2]
print(f"gram centre={centre} P={P} N={N} order={os.environ.get('VMCPDE_GRAM_ORDER','1')} nsplit={os.environ.get('VMCPDE_GRAM_NSPLIT','auto')} ws={need/1e6:.0f}MB: median {ms:.3f} ms, "
      f"{N*P*(P+1)/ms*1e-9:.2f} TFLOP/s  checksum {float(packed[:1000].sum()):.6e}", flush=True)
