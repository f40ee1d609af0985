"""Diagnostic (libvmcpde_stamps.so, built with -DVMC_GRAM_STAMPS): where a Gram main-loop iteration spends its cycles."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vmc_pde_amd import _native as nat
P, N = 4213, 65536
dev = torch.device("cuda", 0)
Pp, ldj = (P + 127) // 128 * 128, N
g = torch.Generator(device=dev).manual_seed(7)
JT = torch.zeros((Pp, ldj), dtype=torch.float64, device=dev)
JT[:P] = torch.randn((P, N), generator=g, dtype=torch.float64, device=dev)
c = JT[:P].mean(1).contiguous()
centre = not (len(sys.argv) > 1 and sys.argv[1] == "nocentre")
packed = torch.empty(nat.lib().vmcpde_gram_packed_doubles(P), dtype=torch.float64, device=dev)
need = nat.lib().vmcpde_gram_workspace_bytes(P, N)
ws = torch.empty(need // 8, dtype=torch.float64, device=dev)
for it in range(3):
    nat.call("vmcpde_gram", nat.ptr(JT), ldj, 0, P, N, nat.ptr(c) if centre else None, None, nat.ptr(packed), nat.ptr(ws), ws.numel() * 8, nat.stream())
torch.cuda.synchronize()
n = 4096 * 4 * 8
buf = (C.c_ulonglong * n)()
rc = nat.lib().vmcpde_debug_read_stamps(buf, n)
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 4, 8).astype(np.float64)
it = a[..., 4]
ok = it > 100
seg = a[..., :4] / np.maximum(it[..., None], 1)
tot = a[..., 5] / np.maximum(it, 1)
print("rc", rc, "workgroups with data", int(ok[:, 0].sum()), "iters", float(np.median(it[ok])))
for w in range(4):
    m = ok[:, w]
    print(f"wave {w}: cycles/iter median: seg0 {np.median(seg[m, w, 0]):.0f}  seg1 {np.median(seg[m, w, 1]):.0f}  seg2 {np.median(seg[m, w, 2]):.0f}  seg3 {np.median(seg[m, w, 3]):.0f}  total {np.median(tot[m, w]):.0f}")
t = tot[ok]
print("centre", centre, "vmcnt wait at stage start, cycles/iter median:", float(np.median((a[..., 6] / np.maximum(it, 1))[ok])))
print("total/iter percentiles 5/25/50/75/95:", np.percentile(t, [5, 25, 50, 75, 95]).round(0))
for i, nm in enumerate(("half1", "stage", "half2", "barrier")):
    v = seg[..., i][ok]
    print(nm, "percentiles 5/50/95:", np.percentile(v, [5, 50, 95]).round(0))
