"""Does a captured HIP graph shorten the Cholesky's chain of ~97 dependent launches?  Direct launches against torch.cuda.CUDAGraph replay
of the same vmcpde_cholesky call (P = 4213 and 16181)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vmc_pde_amd import _native as nat
for P in (4213, 16181):
    Pp = (P + 127) // 128 * 128
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.randn((P, 2 * P), generator=g, dtype=torch.float64, device=dev)
    S = torch.zeros((Pp, Pp), dtype=torch.float64, device=dev)
    S[:P, :P] = X @ X.T / (2 * P)
    del X
    need = nat.lib().vmcpde_chol_workspace_bytes(P)
    ws = torch.empty(need // 8, dtype=torch.float64, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    A = torch.empty_like(S)

    def run():
        nat.call("vmcpde_cholesky", nat.ptr(A), Pp, P, nat.ptr(info), nat.ptr(ws), need, nat.stream())

    def timed(fn, reps=5):
        ts = []
        for _ in range(reps):
            A.copy_(S)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return sorted(ts)[len(ts) // 2]

    A.copy_(S); run(); torch.cuda.synchronize()
    L_direct = A.clone()
    t_direct = timed(run)
    graph = torch.cuda.CUDAGraph()
    A.copy_(S)
    with torch.cuda.graph(graph):
        run()
    t_graph = timed(graph.replay)
    A.copy_(S); graph.replay(); torch.cuda.synchronize()
    same = bool(torch.equal(torch.tril(A[:P, :P]), torch.tril(L_direct[:P, :P])))
    print(f"P={P}: direct {t_direct:.3f} ms, graph replay {t_graph:.3f} ms, identical factor {same}, pinned {int(info.item())}", flush=True)
