"""Round 6, feasibility of VERDICT item 1 BEFORE any new kernel: can the dense Cholesky's latency chain hide behind the S-build on ONE GPU?

The proposal: form tile rows [0, K) of the Gram first (plus the last tile row, which carries the moments), then factor the leading
128K x 128K block on a side stream WHILE the remaining tile rows [K, nb - 1) are still being formed.  This probe runs exactly that with
the entry points that exist (vmcpde_gram_rows, vmcpde_cholesky on a stand-in SPD matrix of the leading block's size) and measures the
wall time of the pair against the two run back to back -- per K, and per way of sharing the chip:
  plain      both on ordinary streams (the factorisation's launches queue behind the Gram's resident workgroups)
  prio       the factorisation on a high-priority stream
  maskN      the last Gram group on a stream masked to (256 - N) CUs, the factorisation on the complementary N CUs
             (hipExtStreamCreateWithCUMask; "i": the N CUs are every (256/N)-th bit, "c": the first N bits)
Reference for what is replaced: tdvp.py:46-52 (S) and :73-94 (solve).  Usage: python tools/diag/pipelined_chol_probe.py [N_samples] [P]
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from vmc_pde_amd import _native as nat  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4155
REPS = int(os.environ.get("REPS", "6"))
dev = torch.device("cuda:0")
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]


def masked_stream(bits):
    """bits: iterable of 256 0/1 -> torch stream restricted to those CUs."""
    words = (C.c_uint32 * 8)()
    for i, b in enumerate(bits):
        if b:
            words[i // 32] |= (1 << (i % 32))
    h = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), 8, words)
    assert rc == 0 and h.value, f"hipExtStreamCreateWithCUMask rc={rc}"
    return torch.cuda.ExternalStream(h.value, device=dev)


def mask_pair(n_res, kind):
    if kind == "i":
        step = 256 // n_res
        res = [1 if i % step == 0 else 0 for i in range(256)]
    else:
        res = [1 if i < n_res else 0 for i in range(256)]
    return masked_stream([1 - b for b in res]), masked_stream(res)


T = nat.GRAM_TILE
Ppad = (P + T - 1) // T * T
nb = Ppad // T
ldj = (N + 15) // 16 * 16
g = torch.Generator(device=dev).manual_seed(7)
JT = torch.zeros((Ppad, ldj), dtype=torch.float64, device=dev)
JT[:P, :N] = torch.randn((P, N), generator=g, dtype=torch.float64, device=dev)
npk = nat.lib().vmcpde_gram_packed_doubles(P)
packed = torch.zeros(npk, dtype=torch.float64, device=dev)
gws = torch.empty(nat.lib().vmcpde_gram_workspace_bytes(P, N) // 8, dtype=torch.float64, device=dev)
# stand-in for the regularised leading block: well conditioned SPD of the full padded size, re-copied before every factorisation
M = torch.randn((Ppad, 2 * Ppad), generator=g, dtype=torch.float64, device=dev)
A0 = M @ M.T / (2 * Ppad)
A0.diagonal().add_(0.05)
del M
A = torch.empty_like(A0)
cws_bytes = nat.lib().vmcpde_chol_workspace_bytes(Ppad)
cws = torch.empty(cws_bytes // 8, dtype=torch.float64, device=dev)
info = torch.zeros(1, dtype=torch.int32, device=dev)


def gram_rows(r0, r1):
    nat.call("vmcpde_gram_rows", nat.ptr(JT), ldj, P, N, r0, r1, nat.ptr(packed), nat.ptr(gws), gws.numel() * 8, nat.stream())


def chol(n):
    A[:n].copy_(A0[:n])
    nat.call("vmcpde_cholesky", nat.ptr(A), Ppad, n, nat.ptr(info), nat.ptr(cws), cws_bytes, nat.stream())


def timed(fn, reps=REPS):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


main = torch.cuda.current_stream(dev)
print(f"N={N} P={P} nb={nb}  (median / min ms over {REPS})", flush=True)
t_full = timed(lambda: gram_rows(0, nb))
print(f"gram all rows, one launch            {t_full[0]:7.3f} / {t_full[1]:7.3f}", flush=True)
t_cfull = timed(lambda: chol(P))
print(f"cholesky full P={P} (+142 MB copy)   {t_cfull[0]:7.3f} / {t_cfull[1]:7.3f}", flush=True)

streams = {"plain": (torch.cuda.Stream(dev), torch.cuda.Stream(dev)),
           "prio": (torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=-1))}
for spec in os.environ.get("MASKS", "32i,32c,64i,64c,16i").split(","):
    spec = spec.strip()
    if spec:
        streams["mask" + spec] = mask_pair(int(spec[:-1]), spec[-1])

for K in [int(v) for v in os.environ.get("KS", "25,27,29").split(",")]:
    nlead = K * T
    tri = lambda r: r * (r + 1) // 2
    share = (tri(nb - 1) - tri(K)) / tri(nb)

    def g1():
        gram_rows(0, K)
        gram_rows(nb - 1, nb)

    def g2():
        gram_rows(K, nb - 1)

    t1, t2, tc = timed(g1), timed(g2), timed(lambda: chol(nlead))
    print(f"K={K}: last group = {share:.1%} of the tiles | G1 {t1[0]:.3f}  G2 {t2[0]:.3f}  chol({nlead}) {tc[0]:.3f}  back to back {t1[0] + t2[0] + tc[0]:.3f}",
          flush=True)

    def seq():
        g1()
        g2()
        chol(nlead)
    ts = timed(seq)
    print(f"    sequential on one stream            {ts[0]:7.3f} / {ts[1]:7.3f}", flush=True)
    for name, (sg, sc) in streams.items():
        def overlapped():
            g1()
            e = torch.cuda.Event()
            e.record(main)
            sg.wait_event(e)
            sc.wait_event(e)
            with torch.cuda.stream(sg):
                g2()
                eg = torch.cuda.Event()
                eg.record(sg)
            with torch.cuda.stream(sc):
                chol(nlead)
                ec = torch.cuda.Event()
                ec.record(sc)
            main.wait_event(eg)
            main.wait_event(ec)
        to = timed(overlapped)
        # and the last group alone on its (possibly masked) stream: what the mask costs the Gram by itself
        def g2_alone():
            e = torch.cuda.Event()
            e.record(main)
            sg.wait_event(e)
            with torch.cuda.stream(sg):
                g2()
                eg = torch.cuda.Event()
                eg.record(sg)
            main.wait_event(eg)

        def c_alone():
            e = torch.cuda.Event()
            e.record(main)
            sc.wait_event(e)
            with torch.cuda.stream(sc):
                chol(nlead)
                ec = torch.cuda.Event()
                ec.record(sc)
            main.wait_event(ec)
        ta, tca = timed(g2_alone, 3), timed(c_alone, 3)
        print(f"    {name:9s} overlapped {to[0]:7.3f} / {to[1]:7.3f}   saves {ts[0] - to[0]:+.3f} of chol {tc[0]:.3f}   "
              f"(G2 alone on its stream {ta[0]:.3f}, chol alone on its stream {tca[0]:.3f})", flush=True)
assert int(info.item()) == 0
