"""Round 5: is rocSOLVER's syevd on the r x r core of the spectral engine (4.35 ms at r = 206) bound by the HOST's launch rate?
(a) eigh timed from an idle stream (every launch waits for the host); (b) the same call queued behind a long kernel, so that all its
launches are in the queue before the first can run -- HIP events around it then give the device-side time of the launch chain."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
big = torch.randn(8192, 8192, device=dev, dtype=torch.float32, generator=g)
for r in (206, 256, 128, 512):
    A = torch.randn(r, r, device=dev, dtype=torch.float64, generator=g)
    T = A @ A.T / r
    # graded spectrum like the model's core (seven decades)
    w = torch.logspace(0, -7, r, device=dev, dtype=torch.float64)
    Q, _ = torch.linalg.qr(A)
    T = (Q * w) @ Q.T
    for _ in range(3):
        torch.linalg.eigh(T)
    torch.cuda.synchronize()
    ta = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        torch.linalg.eigh(T)
        torch.cuda.synchronize()
        ta.append((time.perf_counter() - t0) * 1e3)
    tb, th = [], []
    for _ in range(5):
        torch.cuda.synchronize()
        for _ in range(6):
            big @ big                      # ~25 ms of queued work in front
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        torch.linalg.eigh(T)
        e1.record()
        th.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        tb.append(e0.elapsed_time(e1))
    print(f"r={r}: eigh from an idle stream {sorted(ta)[2]:.2f} ms wall; queued behind work: device span {sorted(tb)[2]:.2f} ms, host enqueue {sorted(th)[2]:.2f} ms", flush=True)
