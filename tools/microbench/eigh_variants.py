"""Eigensolvers for the spectral solve (tdvp.py:57-64) at the reference's sizes and at C2: host LAPACK (numpy, thread-capped),
rocSOLVER syevd (torch.linalg.eigh) and rocSOLVER's Jacobi syevj / divide-and-conquer+Jacobi syevdj called through ctypes.
S is a rank-deficient Gram matrix like the real one (rank ~ 0.4 P)."""
import ctypes as C, sys, time
import numpy as np, torch
from threadpoolctl import threadpool_limits

dev = torch.device("cuda", 0)
rs = C.CDLL("librocsolver.so")
rb = C.CDLL("librocblas.so")
handle = C.c_void_p()
assert rb.rocblas_create_handle(C.byref(handle)) == 0
rb.rocblas_set_stream(handle, C.c_void_p(torch.cuda.current_stream().cuda_stream))
ESORT_ASC, EVECT_ORIG, FILL_LOWER = 252, 211, 122

def gpu_time(f, reps=5):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3

for P in [int(a) for a in sys.argv[1:]] or [392, 411, 1024, 4213]:
    g = torch.Generator(device=dev).manual_seed(P)
    X = torch.randn((P, int(0.4 * P)), generator=g, dtype=torch.float64, device=dev)
    S = X @ X.T / X.shape[1]
    Sh = S.cpu().numpy()
    with threadpool_limits(limits=1 if P < 600 else (8 if P < 2000 else 16)):
        np.linalg.eigh(Sh); t = time.perf_counter(); ev_h, V_h = np.linalg.eigh(Sh); t_host = (time.perf_counter() - t) * 1e3
    t_syevd = gpu_time(lambda: torch.linalg.eigh(S), 3)
    W = torch.empty(P, dtype=torch.float64, device=dev)
    resid = torch.empty(1, dtype=torch.float64, device=dev)
    nsw = torch.empty(1, dtype=torch.int32, device=dev)
    info = torch.empty(1, dtype=torch.int32, device=dev)
    out = {}
    def syevj():
        A = S.clone()
        rc = rs.rocsolver_dsyevj(handle, ESORT_ASC, EVECT_ORIG, FILL_LOWER, P, C.c_void_p(A.data_ptr()), P, C.c_double(0.0), C.c_void_p(resid.data_ptr()),
                                 100, C.c_void_p(nsw.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(info.data_ptr()))
        assert rc == 0, rc
        out["A"] = A
    t_syevj = gpu_time(syevj, 3)
    V = out["A"].T.contiguous()           # column-major eigenvectors
    err = float((S @ V - V * W).abs().max() / S.abs().max())
    orth = float((V.T @ V - torch.eye(P, dtype=torch.float64, device=dev)).abs().max())
    evd = float(np.abs(W.cpu().numpy() - ev_h).max() / ev_h[-1])
    line = f"P={P}: host LAPACK {t_host:8.2f} ms | rocSOLVER syevd {t_syevd:8.2f} ms | syevj {t_syevj:8.2f} ms (sweeps {int(nsw)}, info {int(info)}, |SV-VW| {err:.1e}, |VtV-I| {orth:.1e}, ev diff {evd:.1e})"
    if hasattr(rs, "rocsolver_dsyevdj"):
        def syevdj():
            A = S.clone()
            rc = rs.rocsolver_dsyevdj(handle, EVECT_ORIG, FILL_LOWER, P, C.c_void_p(A.data_ptr()), P, C.c_void_p(W.data_ptr()), C.c_void_p(info.data_ptr()))
            assert rc == 0, rc
            out["A"] = A
        t_dj = gpu_time(syevdj, 3)
        V = out["A"].T.contiguous()
        line += f" | syevdj {t_dj:8.2f} ms (|SV-VW| {float((S @ V - V * W).abs().max() / S.abs().max()):.1e})"
    print(line, flush=True)
