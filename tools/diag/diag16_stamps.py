"""Stage timeline of chol_diag16_body (build: python -m vmc_pde_amd.build --variant stamps -DVMC_DIAG16_STAMPS)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["VMCPDE_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "vmc_pde_amd", "_lib", "libvmcpde_stamps.so")
import torch
from vmc_pde_amd import _native as nat
P = 128
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
X = torch.randn((P, 2 * P), generator=g, dtype=torch.float64, device=dev)
S = X @ X.T / (2 * P)
need = nat.lib().vmcpde_chol_workspace_bytes(P)
ws = torch.empty(need // 8, dtype=torch.float64, device=dev)
info = torch.zeros(1, dtype=torch.int32, device=dev)
for it in range(3):
    A = S.clone()
    nat.call("vmcpde_cholesky", nat.ptr(A), P, P, nat.ptr(info), nat.ptr(ws), need, nat.stream())
    torch.cuda.synchronize()
buf = (C.c_longlong * 64)()
nat.lib().vmcpde_debug_read_diag16_stamps(buf)
st = list(buf)
t0 = st[0]
print("load -> first sweep start: (stamp 0 is after the load barrier)")
print(f"sweep(0): {(st[1]-st[0])*10} ns")
for jb in range(8):
    b = 2 + 4 * jb
    s2 = st[b] - (st[1] if jb == 0 else st[b - 1])
    line = f"step {jb}: S2+barrier {s2*10} ns"
    if jb < 7:
        line += f", S3a {(st[b+1]-st[b])*10} ns, sweep {(st[b+2]-st[b+1])*10} ns, join barrier {(st[b+3]-st[b+2])*10} ns"
    else:
        line += f", tail to barrier {(st[b+3]-st[b])*10} ns"
    print(line)
print(f"total steps {(st[40]-st[0])*10} ns, Linv store {(st[41]-st[40])*10} ns")
print(f"wave 1, step 0: S3b ({9} blocks) {(st[51]-st[50])*10} ns, S3c {(st[52]-st[51])*10} ns, L write-back {(st[53]-st[52])*10} ns; started {(st[50]-st[2])*10} ns after the S2 barrier")
