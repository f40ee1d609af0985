"""Round 5: what kind of phase BETWEEN two S-builds makes the next S-build slower?  (The Gram takes 17.1 ms alone, 17.25 in the
Cholesky loop with ~1.7 ms of light kernels between two Grams, 18.4 ms in the spectral loop with ~7.5 ms.)  Loop: Gram, filler; the
Gram's HIP-event bracket per filler kind and length."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vmc_pde_amd import _native as nat
P, N = 4213, 65536
dev = torch.device("cuda", 0)
Pp = (P + 127) // 128 * 128
g = torch.Generator(device=dev).manual_seed(7)
JT = torch.zeros((Pp, N), dtype=torch.float64, device=dev)
JT[:P] = torch.randn((P, N), generator=g, dtype=torch.float64, device=dev)
packed = torch.empty(nat.lib().vmcpde_gram_packed_doubles(P), dtype=torch.float64, device=dev)
need = nat.lib().vmcpde_gram_workspace_bytes(P, N)
ws = torch.empty(need // 8, dtype=torch.float64, device=dev)
small = torch.zeros(256, device=dev)
stream_buf = torch.zeros(64 * 1024 * 1024, device=dev)          # 256 MB: a full-chip memory-bound filler
T206 = torch.randn(206, 206, device=dev, dtype=torch.float64)
T206 = T206 @ T206.T


def gram():
    nat.call("vmcpde_gram", nat.ptr(JT), N, 0, P, N, None, None, nat.ptr(packed), nat.ptr(ws), ws.numel() * 8, nat.stream())


def filler(kind, ms):
    if kind == "none":
        return
    if kind == "idle":                     # the device idles: the host waits, then sleeps
        torch.cuda.synchronize()
        time.sleep(ms * 1e-3)
    elif kind == "tiny":                   # a chain of tiny dependent launches (~4.5 us each incl. the boundary)
        for _ in range(int(ms * 1e3 / 4.5)):
            small.add_(1.0)
    elif kind == "onewave":                # ONE wave spins (torch.cuda._sleep counts ~cycles)
        torch.cuda._sleep(int(ms * 1e-3 * 2.0e9))
    elif kind == "stream":                 # every CU busy, memory-bound, little power
        for _ in range(max(1, int(ms / 0.11))):
            stream_buf.add_(1.0)
    elif kind == "eigh":
        for _ in range(max(1, int(round(ms / 4.45)))):
            torch.linalg.eigh(T206)


for kind, ms in [("none", 0), ("idle", 2), ("idle", 8), ("tiny", 2), ("tiny", 8), ("onewave", 2), ("onewave", 8), ("stream", 2), ("stream", 8),
                 ("eigh", 4.45), ("eigh", 8.9), ("none", 0)]:
    ts, fs = [], []
    for it in range(14):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        gram()
        e1.record()
        filler(kind, ms)
        e2.record()
        if it >= 4:
            ts.append((e0, e1))
            fs.append((e1, e2))
    torch.cuda.synchronize()
    tg = sorted(a.elapsed_time(b) for a, b in ts)
    tf = sorted(a.elapsed_time(b) for a, b in fs)
    print(f"filler {kind:8s} {ms:5.2f} ms (measured {tf[len(tf) // 2]:6.2f}): gram median {tg[len(tg) // 2]:.3f} ms  min {tg[0]:.3f}  max {tg[-1]:.3f}", flush=True)
