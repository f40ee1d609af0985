"""Round-3 experiment (verdict item 6): the error-free sliced Gram on the INT8 matrix pipe (csrc/gram_i8.hip) against the fp64 MFMA
syrk on the model-free micro-benchmark -- random J ~ N(0,1), 65 536 samples x 4096 parameters, seed 7 -- INCLUDING the
exponent and slicing passes.  Accept only if it beats 16.5 ms with rel-Frobenius <= 1e-12 against the fp64 kernel and is bit-exact
on integer data.
    python tools/probe_gram_i8.py [P] [N] [reps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmc_pde_amd import _native as nat

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda", 0)
Pp, ldj = (P + 127) // 128 * 128, (N + 15) // 16 * 16
lib = nat.lib()
npk = lib.vmcpde_gram_packed_doubles(P)
nb = Pp // 128


def lower_mask():
    """entries of the packed tiles that finalize reads: everything in off-diagonal tiles, row >= col in diagonal ones"""
    m = torch.ones((npk // 16384, 128, 128), dtype=torch.bool, device=dev)
    tri = torch.tril(torch.ones((128, 128), dtype=torch.bool, device=dev))
    t = 0
    for ti in range(nb):
        for tj in range(ti + 1):
            if ti == tj:
                m[t] = tri
            t += 1
    return m.reshape(-1)


def run_f64(JT):
    packed = torch.empty(npk, dtype=torch.float64, device=dev)
    need = lib.vmcpde_gram_workspace_bytes(P, N)
    ws = torch.empty(max(need // 8, 1), dtype=torch.float64, device=dev)
    ts = []
    for it in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        nat.call("vmcpde_gram", nat.ptr(JT), ldj, 0, P, N, None, None, nat.ptr(packed), nat.ptr(ws), ws.numel() * 8, nat.stream())
        e1.record()
        torch.cuda.synchronize()
        if it:
            ts.append(e0.elapsed_time(e1))
    return packed, sorted(ts)[len(ts) // 2]


STATS = torch.zeros(1, dtype=torch.float64, device=dev)


def run_i8(JT, cfg):
    ns, bits = cfg
    packed = torch.empty(npk, dtype=torch.float64, device=dev)
    need = lib.vmcpde_gram_i8_workspace_bytes(P, N, ns, bits)
    ws = torch.empty(need // 8 + 1, dtype=torch.float64, device=dev)
    ts = []
    for it in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        nat.call("vmcpde_gram_i8", nat.ptr(JT), ldj if JT.dtype == torch.float64 else JT.shape[1], 1 if JT.dtype == torch.float32 else 0, P, N, ns, bits, None, None,
                 nat.ptr(packed), nat.ptr(STATS), nat.ptr(ws), ws.numel() * 8, nat.stream())
        e1.record()
        torch.cuda.synchronize()
        if it:
            ts.append(e0.elapsed_time(e1))
    return packed, sorted(ts)[len(ts) // 2]


mask = lower_mask()
CFGS = ((6, 8), (6, 7), (7, 7), (4, 8))        # (slices, bits per digit)
g = torch.Generator(device=dev).manual_seed(7)
JT = torch.zeros((Pp, ldj), dtype=torch.float64, device=dev)
JT[:P, :N] = torch.randn((P, N), generator=g, dtype=torch.float64, device=dev)
res = {"P": P, "N": N, "workload": "random J ~ N(0,1), seed 7 (model-free micro-benchmark, SURVEY 8d)"}
ref, ms64 = run_f64(JT)
res["fp64_mfma_ms"] = ms64
print(f"fp64 MFMA syrk: {ms64:.3f} ms ({N * P * (P + 1) / ms64 * 1e-9:.1f} TFLOP/s)", flush=True)
for ns in CFGS:
    got, ms = run_i8(JT, ns)
    err = float(torch.linalg.norm((got - ref)[mask]) / torch.linalg.norm(ref[mask]))
    worst = float(((got - ref)[mask]).abs().max() / ref[mask].abs().max())
    res[f"i8_{ns[0]}x{ns[1]}bit"] = {"ms_incl_slicing": ms, "rel_frobenius_vs_fp64": err, "max_abs_over_max": worst,
                             "pairs": ns[0] * (ns[0] + 1) // 2, "int8_ops": ns[0] * (ns[0] + 1) // 2 * N * P * (P + 1)}
    print(f"int8 sliced, {ns[0]} slices x {ns[1]} bits ({ns[0] * (ns[0] + 1) // 2} pairs): {ms:.3f} ms incl. exponent + slicing passes, rel-Frobenius vs fp64 {err:.2e}, "
          f"max |d| / max |S| {worst:.2e}, {ns[0] * (ns[0] + 1) // 2 * N * P * (P + 1) / ms * 1e-12:.2f} POP/s overall, row max/rms figure {float(STATS[0]):.1f}", flush=True)
# exactness on integer data (both kernels are exact there)
Ji = torch.zeros((Pp, ldj), dtype=torch.float64, device=dev)
Ji[:P, :N] = torch.randint(-7, 8, (P, N), generator=g, device=dev).double()
ref_i, _ = run_f64(Ji)
for ns in CFGS:
    got_i, _ = run_i8(Ji, ns)
    same = bool(torch.equal(got_i[mask], ref_i[mask]))
    res[f"i8_{ns[0]}x{ns[1]}bit"]["bit_exact_on_integer_J"] = same
    print(f"integer J, {ns[0]} slices x {ns[1]} bits: bit-exact = {same}", flush=True)
# a heavy-tailed Jacobian (row max / rms ~ 40): what the fixed number of slices costs there
Jh = torch.zeros((Pp, ldj), dtype=torch.float64, device=dev)
Jh[:P, :N] = torch.randn((P, N), generator=g, dtype=torch.float64, device=dev) ** 3
ref_h, _ = run_f64(Jh)
for ns in CFGS:
    got_h, _ = run_i8(Jh, ns)
    err = float(torch.linalg.norm((got_h - ref_h)[mask]) / torch.linalg.norm(ref_h[mask]))
    res[f"i8_{ns[0]}x{ns[1]}bit"]["rel_frobenius_heavy_tailed_J"] = err
    print(f"heavy-tailed J (cubed normals), {ns[0]} slices x {ns[1]} bits: rel-Frobenius vs fp64 {err:.2e}, row max/rms figure {float(STATS[0]):.1f}", flush=True)
# BASELINE config 5: an fp32 Jacobian, four 8-bit digits (31 bits) against the widening fp64 kernel on the same fp32 values
J32 = torch.zeros((Pp, ldj), dtype=torch.float32, device=dev)
J32[:P, :N] = JT[:P, :N].float()
packed32 = torch.empty(npk, dtype=torch.float64, device=dev)
need = lib.vmcpde_gram_workspace_bytes(P, N)
ws32 = torch.empty(max(need // 8, 1), dtype=torch.float64, device=dev)
ts = []
for it in range(reps + 1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    nat.call("vmcpde_gram", nat.ptr(J32), ldj, 1, P, N, None, None, nat.ptr(packed32), nat.ptr(ws32), ws32.numel() * 8, nat.stream())
    e1.record()
    torch.cuda.synchronize()
    if it:
        ts.append(e0.elapsed_time(e1))
ms_w = sorted(ts)[len(ts) // 2]
got32, ms32 = run_i8(J32, (4, 8))
err32 = float(torch.linalg.norm((got32 - packed32)[mask]) / torch.linalg.norm(packed32[mask]))
err64 = float(torch.linalg.norm((got32 - ref)[mask]) / torch.linalg.norm(ref[mask]))
res["i8_4x8bit_on_fp32_J"] = {"ms_incl_slicing": ms32, "widening_fp64_kernel_ms": ms_w, "rel_frobenius_vs_widening_fp64_kernel": err32,
                              "rel_frobenius_vs_fp64_J": err64}
print(f"fp32 Jacobian (config 5): int8 4 x 8 bits {ms32:.3f} ms incl. slicing against {ms_w:.3f} ms for the widening fp64 kernel; rel-Frobenius vs that kernel "
      f"{err32:.2e}, vs the fp64 Jacobian's Gram {err64:.2e}", flush=True)
print(json.dumps(res))
