"""Round 5: the r x r core of the spectral engine -- own tridiagonalisation (csrc/tridiag.hip) + rocSOLVER stedc + own back-transformation
against torch.linalg.eigh (rocSOLVER syevd), HIP-event times of each piece on graded matrices of the core's size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import vmc_pde_amd as V
from vmc_pde_amd import _native as nat, rocsolver


def timed(f, reps=10):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


for n in (64, 128, 206, 256):
    g = torch.Generator().manual_seed(n)
    Q0, _ = torch.linalg.qr(torch.randn(n, n, generator=g, dtype=torch.float64))
    lam = 2.0e4 * 10.0 ** (-torch.arange(n, dtype=torch.float64) / (n / 7.0))
    A = ((Q0 * lam) @ Q0.T).cuda()
    A = 0.5 * (A + A.T)
    tp = V.TDVP()
    w = {}
    tp._core_eigh_tridiag(w, A, n, A.device)
    td = w["td"]
    t_all = timed(lambda: tp._core_eigh_tridiag(w, A, n, A.device))
    t_lib = timed(lambda: torch.linalg.eigh(A))
    t_sy = timed(lambda: nat.call("vmcpde_sytrd_small", nat.ptr(A), n, n, nat.ptr(td["D"]), nat.ptr(td["E"]), nat.ptr(td["tau"]), nat.ptr(td["V"]), 256, nat.stream()))
    Z = td["Z"][:n * n].view(n, n)
    D0, E0 = td["D"].clone(), td["E"].clone()

    def st():
        td["D"].copy_(D0); td["E"].copy_(E0)
        rocsolver.stedc(td["D"][:n], td["E"][:n], Z, td["info"])
    t_st = timed(st)
    t_bt = timed(lambda: nat.call("vmcpde_apply_reflectors", nat.ptr(td["V"]), 256, nat.ptr(td["tau"]), n, nat.ptr(Z), n, n, nat.stream()))
    print(f"n {n}: library eigh {t_lib:.3f} ms | own route {t_all:.3f} ms = sytrd {t_sy:.3f} + stedc {t_st:.3f} (incl. two small copies) + back-transformation {t_bt:.3f}", flush=True)
