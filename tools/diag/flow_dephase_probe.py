"""Round 6: can flow_eval's compute phase (Taylor pass: no stores) and its store phase (reverse sweep: 2.2 GB of Jacobian rows)
overlap ACROSS workgroups?  All 1024 workgroups of the C2 launch are resident at once and walk the two phases in lockstep.  This probe evaluates the
two halves of the batch as two launches on two streams (the C ABI takes offset pointers: the same bytes land in the same Jacobian buffer) and delays
the second by a spin of d microseconds, so that its Taylor pass runs under the first half's stores.  Printed: wall time from the first launch to the
end of the second, per delay -- against the single launch.  What it cannot show is a FREE stagger: the delayed half leaves its share of every CU idle
for d at the start and the first half does the same at the end, so only a gain larger than d would pay.  Usage: python tools/diag/flow_dephase_probe.py [C2|C4]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from vmc_pde_amd import _native as nat  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
dev = torch.device("cuda", 0)
vs, eq, tdvp, theta = bench.build_problem(dev, wl)
N = bench.WORKLOADS[wl]["n_samp"]
x, _ = vs.sample(N)
x = x.reshape(-1, vs.dim).contiguous()
P, d = vs.numParameters, vs.dim
ws = vs._workspace(N)
JT, ldj = ws["JT"], ws["ldj"]
dirs_np = eq._dirs(vs) if hasattr(eq, "_dirs") else None
# the Hessian directions of the Kramers operator: the momentum axes (evolutionEq.py:107-119)
import numpy as np  # noqa: E402
dirs = torch.from_numpy(np.eye(d)[1::2].copy()).to(dev)
D = dirs.shape[0]
logp = torch.empty(N, dtype=torch.float64, device=dev)
g = torch.empty((N, d), dtype=torch.float64, device=dev)
hv = torch.empty((N, D), dtype=torch.float64, device=dev)
th = vs._theta


def flow(n0, n1):
    n = n1 - n0
    nat.call("vmcpde_flow_eval", vs.net.handle, nat.ptr(th), nat.C.c_void_p(x.data_ptr() + n0 * d * 8), n, nat.ptr(dirs), D,
             nat.C.c_void_p(logp.data_ptr() + n0 * 8), nat.C.c_void_p(g.data_ptr() + n0 * d * 8), nat.C.c_void_p(hv.data_ptr() + n0 * D * 8),
             nat.C.c_void_p(JT.data_ptr() + n0 * 8), ldj, 0, None, 0, nat.stream())


def timed(fn, reps=7):
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
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
print(f"{wl}: N={N} P={P}", flush=True)
t = timed(lambda: flow(0, N))
print(f"one launch, all samples                     {t[0]:.3f} / {t[1]:.3f} ms", flush=True)
ref = JT[:P, :N].clone()
t = timed(lambda: flow(0, N // 2))
print(f"first half alone                            {t[0]:.3f} / {t[1]:.3f} ms", flush=True)
t = timed(lambda: (flow(0, N // 2), flow(N // 2, N)))
print(f"two halves back to back on one stream       {t[0]:.3f} / {t[1]:.3f} ms", flush=True)
cycles_per_us = 100            # torch.cuda._sleep counts the 100 MHz wall clock on this stack? calibrated below
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(1000000); e1.record(); e1.synchronize()
cycles_per_us = 1000000 / (e0.elapsed_time(e1) * 1e3)
print(f"(sleep calibration: {cycles_per_us:.1f} cycles per microsecond)", flush=True)
for delay_us in (0, 50, 100, 150, 200, 300):
    def staggered():
        e = torch.cuda.Event()
        e.record(main)
        s1.wait_event(e)
        s2.wait_event(e)
        with torch.cuda.stream(s1):
            flow(0, N // 2)
            a = torch.cuda.Event()
            a.record(s1)
        with torch.cuda.stream(s2):
            if delay_us:
                torch.cuda._sleep(int(delay_us * cycles_per_us))
            flow(N // 2, N)
            b = torch.cuda.Event()
            b.record(s2)
        main.wait_event(a)
        main.wait_event(b)
    t = timed(staggered)
    print(f"two streams, second half delayed {delay_us:3d} us   {t[0]:.3f} / {t[1]:.3f} ms   (minus the delay: {t[0] - delay_us * 1e-3:.3f})", flush=True)
assert torch.equal(JT[:P, :N], ref), "the split launches must write the same Jacobian"
print("split launches wrote the same Jacobian bits", flush=True)
