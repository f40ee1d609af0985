"""Round 5: the Householder chain of the rank-revealing engine (vmcpde_householder_rows) on a graded P x r factor: sequential pairs
(VMCPDE_HQR_BATCHED=0), batched pairs and batched fours -- time, and agreement of (Y, tau, R) between the forms."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from vmc_pde_amd import _native as nat
for P, r in ((4213, 206), (2000, 131), (8000, 206), (16181, 262)):
    ldl = (P + 63) // 64 * 64
    g = torch.Generator(device="cuda").manual_seed(P)
    L0 = torch.zeros((r, ldl), dtype=torch.float64, device="cuda")
    L0[:, :P] = torch.randn((r, P), generator=g, dtype=torch.float64, device="cuda") * torch.logspace(0, -7, r, dtype=torch.float64, device="cuda")[:, None]
    outs = {}
    for name, env in (("sequential pairs", {"VMCPDE_HQR_BATCHED": "0"}), ("batched pairs", {"VMCPDE_HQR_BLOCK": "2"}), ("fours (batched up to P = 6145)", {})):
        for k in ("VMCPDE_HQR_BATCHED", "VMCPDE_HQR_BLOCK"):
            os.environ.pop(k, None)
        os.environ.update(env)
        YT = torch.empty((r, ldl), dtype=torch.float64, device="cuda")
        tau = torch.empty(r, dtype=torch.float64, device="cuda")
        hq = torch.empty(5 * r + 5, dtype=torch.float64, device="cuda")
        ts = []
        for it in range(6):
            LT = L0.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            nat.call("vmcpde_householder_rows", nat.ptr(LT), ldl, P, r, nat.ptr(YT), nat.ptr(tau), nat.ptr(hq), nat.stream())
            e1.record(); torch.cuda.synchronize()
            if it: ts.append(e0.elapsed_time(e1))
        outs[name] = (sorted(ts)[len(ts) // 2], YT.clone(), tau.clone(), torch.triu(LT[:, :r].T.clone(), 1))      # (the strict upper part: the diagonal of R is not stored)
    ref = outs["sequential pairs"]
    msg = []
    for name, (t, Y, ta, R) in outs.items():
        msg.append(f"{name} {t:.3f} ms (|dY| {float((Y - ref[1]).abs().max()):.1e}, |dtau| {float((ta - ref[2]).abs().max()):.1e}, |dR|/|R| {float((R - ref[3]).abs().max() / ref[3].abs().max()):.1e})")
    print(f"P {P} r {r}: " + "; ".join(msg), flush=True)
