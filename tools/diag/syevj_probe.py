"""The r x r core of the rank-revealing spectral engine: rocSOLVER's syevd (what torch.linalg.eigh runs) against its Jacobi solver
syevj called through ctypes -- cold, and warm-started in the previous call's eigenbasis (a nearly diagonal matrix converges in a few
sweeps) -- on graded symmetric positive definite matrices of the size and spectrum of Q1^T S Q1 at the headline workload."""
import ctypes as C, os, sys, time
import numpy as np, torch
dev = torch.device("cuda", 0)
lib = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librocsolver.so"))
blas = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librocblas.so"))
h = C.c_void_p()
assert blas.rocblas_create_handle(C.byref(h)) == 0
assert blas.rocblas_set_stream(h, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
lib.rocsolver_dsyevj.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_void_p,
                                 C.c_void_p, C.c_void_p]
ESORT_ASC, EVECT_ORIG, FILL_UPPER = 252, 211, 121


def syevj(T, max_sweeps=30, abstol=0.0):
    n = T.shape[0]
    A = T.clone().contiguous()            # (symmetric: row- and column-major agree); overwritten with the eigenvectors (columns, column-major)
    W = torch.empty(n, dtype=torch.float64, device=dev)
    resid = torch.zeros(1, dtype=torch.float64, device=dev)
    nsw = torch.zeros(1, dtype=torch.int32, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    rc = lib.rocsolver_dsyevj(h, ESORT_ASC, EVECT_ORIG, FILL_UPPER, n, C.c_void_p(A.data_ptr()), n, abstol, C.c_void_p(resid.data_ptr()), max_sweeps,
                              C.c_void_p(nsw.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(info.data_ptr()))
    assert rc == 0, rc
    return W, A.T, nsw, info              # A holds V column-major = V^T row-major


def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = f()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps, out


for r in (206, 288):
    g = torch.Generator().manual_seed(r)
    lam = 2.0e4 * 10.0 ** (-torch.arange(r, dtype=torch.float64) / 15.0)
    Q, _ = torch.linalg.qr(torch.randn(r, r, generator=g, dtype=torch.float64))
    T0 = ((Q * lam) @ Q.T).to(dev)
    T0 = 0.5 * (T0 + T0.T)
    E = torch.randn(r, r, generator=g, dtype=torch.float64).to(dev)
    T1 = T0 + 1e-3 * (T0 @ E + E.T @ T0) * 0.5 / np.sqrt(r)      # the next call's matrix: a relative change of 1e-3
    T1 = 0.5 * (T1 + T1.T)
    t_d, (w_d, V_d) = timed(lambda: torch.linalg.eigh(T1))
    t_j, (w_j, V_j, nsw, info) = timed(lambda: syevj(T1))
    _, W0 = torch.linalg.eigh(T0)

    def warm():
        Tw = W0.T @ T1 @ W0
        w, V, n2, i2 = syevj(0.5 * (Tw + Tw.T))
        return w, W0 @ V, n2, i2
    t_w, (w_w, V_w, nsw_w, info_w) = timed(warm)
    def err(w, V):
        return (float((w - w_d).abs().max() / w_d[-1]), float(torch.linalg.norm(T1 @ V - V * w) / w_d[-1]), float((V.T @ V - torch.eye(r, dtype=torch.float64, device=dev)).abs().max()))
    print(f"r {r}: syevd {t_d:.2f} ms | syevj cold {t_j:.2f} ms ({int(nsw)} sweeps, info {int(info)}) err {err(w_j, V_j)} | "
          f"syevj warm {t_w:.2f} ms ({int(nsw_w)} sweeps) err {err(w_w, V_w)}", flush=True)
