"""Timing probe: bordered Cholesky + back substitution at P (default 4213); VMCPDE_BACKSOLVE_PERSISTENT=0/1."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmc_pde_amd import _native as nat
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4213
Pp = (P + 1 + 127) // 128 * 128
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
X = torch.randn((P, 2 * P), generator=g, dtype=torch.float64, device=dev)
S = X @ X.T / (2 * P)
F = torch.randn(P, generator=g, dtype=torch.float64, device=dev)
need = nat.lib().vmcpde_chol_workspace_bytes(P + 1)
ws = torch.empty(need // 8, dtype=torch.float64, device=dev)
info = torch.zeros(1, dtype=torch.int32, device=dev)
u = torch.empty(P, dtype=torch.float64, device=dev)
for it in range(5):
    A = torch.zeros((Pp, Pp), dtype=torch.float64, device=dev)
    A[:P, :P] = S
    A[P, :P] = F
    A[P, P] = 1e200
    nat.call("vmcpde_cholesky", nat.ptr(A), Pp, P + 1, nat.ptr(info), nat.ptr(ws), need, nat.stream())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    nat.call("vmcpde_chol_backsolve", nat.ptr(A), Pp, P, nat.ptr(A[P]), nat.ptr(u), nat.ptr(ws), need, nat.stream())
    e1.record()
    torch.cuda.synchronize()
    res = float(torch.linalg.norm(S @ u - F) / torch.linalg.norm(F))
    print(f"backsolve P={P} persistent={os.environ.get('VMCPDE_BACKSOLVE_PERSISTENT', '1')}: {e0.elapsed_time(e1):.3f} ms, residual {res:.2e}", flush=True)
