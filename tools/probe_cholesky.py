"""Timing probe: blocked Cholesky of a random SPD matrix (P = 4213) -- used with rocprofv3 / VMCPDE_DIAG_PHASE."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmc_pde_amd import _native as nat
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4213
Pp = (P + 127) // 128 * 128
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
X = torch.randn((P, 2 * P), generator=g, dtype=torch.float64, device=dev)
S = X @ X.T / (2 * P)
need = nat.lib().vmcpde_chol_workspace_bytes(P)
ws = torch.empty(need // 8, dtype=torch.float64, device=dev)
info = torch.zeros(1, dtype=torch.int32, device=dev)
for it in range(4):
    A = torch.zeros((Pp, Pp), dtype=torch.float64, device=dev)
    A[:P, :P] = S
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    nat.call("vmcpde_cholesky", nat.ptr(A), Pp, P, nat.ptr(info), nat.ptr(ws), need, nat.stream())
    e1.record()
    torch.cuda.synchronize()
    print(f"cholesky P={P}: {e0.elapsed_time(e1):.3f} ms, pinned={int(info.item())}", flush=True)
