"""Round 5: what would the r x r core cost with an own tridiagonalisation?  rocSOLVER's pieces timed alone through ctypes on a graded matrix
of the core's size: sytrd (the part an own single-workgroup kernel would replace), stedc (tridiagonal divide & conquer, kept), ormtr."""
import ctypes as C, os, time
import numpy as np, scipy.linalg, torch
dev = torch.device("cuda", 0)
lib = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librocsolver.so"))
blas = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librocblas.so"))
h = C.c_void_p()
assert blas.rocblas_create_handle(C.byref(h)) == 0
assert blas.rocblas_set_stream(h, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
EVECT_TRIDIAG, EVECT_ORIG, FILL_LOWER = 212, 211, 122
lib.rocsolver_dstedc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
lib.rocsolver_dstedcj.argtypes = lib.rocsolver_dstedc.argtypes
lib.rocsolver_dsytrd.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]


def timed(f, reps=8):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2], out


for r in (128, 206, 256, 288):
    g = torch.Generator().manual_seed(r)
    lam = 2.0e4 * 10.0 ** (-torch.arange(r, dtype=torch.float64) / 15.0)
    Q, _ = torch.linalg.qr(torch.randn(r, r, generator=g, dtype=torch.float64))
    T = (Q * lam) @ Q.T
    T = 0.5 * (T + T.T)
    Hs = scipy.linalg.hessenberg(T.numpy())
    d0, e0 = np.diag(Hs).copy(), np.diag(Hs, -1).copy()
    D, E = torch.empty(r, dtype=torch.float64, device=dev), torch.empty(r, dtype=torch.float64, device=dev)
    Z = torch.empty((r, r), dtype=torch.float64, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    d0t, e0t = torch.from_numpy(d0).to(dev), torch.from_numpy(np.append(e0, 0.0)).to(dev)

    def stedc():
        D.copy_(d0t); E.copy_(e0t)
        rc = lib.rocsolver_dstedc(h, EVECT_TRIDIAG, r, C.c_void_p(D.data_ptr()), C.c_void_p(E.data_ptr()), C.c_void_p(Z.data_ptr()), r, C.c_void_p(info.data_ptr()))
        assert rc == 0, rc
        return D.clone(), Z.clone()
    t_s, (w, Zc) = timed(stedc)

    def stedcj():
        D.copy_(d0t); E.copy_(e0t)
        rc = lib.rocsolver_dstedcj(h, EVECT_TRIDIAG, r, C.c_void_p(D.data_ptr()), C.c_void_p(E.data_ptr()), C.c_void_p(Z.data_ptr()), r, C.c_void_p(info.data_ptr()))
        assert rc == 0, rc
        return D.clone(), Z.clone()
    try:
        t_j, (wj, Zj) = timed(stedcj)
        Znj = Zj.cpu().numpy().T
        print(f"r {r}: stedcj alone {t_j:.2f} ms (|dw| {np.abs(np.sort(wj.cpu().numpy()) - w_ref0).max() if False else 0:.0f}; Z^T Z - I {np.abs(Znj.T @ Znj - np.eye(r)).max():.1e}, info {int(info)})", flush=True)
    except Exception as exc:
        print("stedcj failed:", exc, flush=True)
    Ttri = np.diag(d0) + np.diag(e0, -1) + np.diag(e0, 1)
    w_ref = np.linalg.eigvalsh(Ttri)
    Zn = Zc.cpu().numpy().T                       # column-major on the device
    wn = w.cpu().numpy()
    order = np.argsort(wn)
    err_w = np.abs(wn[order] - w_ref).max() / w_ref[-1]
    res = np.linalg.norm(Ttri @ Zn - Zn * wn) / w_ref[-1]
    orth = np.abs(Zn.T @ Zn - np.eye(r)).max()
    A = T.to(dev).contiguous()
    Dd, Ee, tau = (torch.empty(r, dtype=torch.float64, device=dev) for _ in range(3))

    def sytrd():
        A2 = A.clone()
        rc = lib.rocsolver_dsytrd(h, FILL_LOWER, r, C.c_void_p(A2.data_ptr()), r, C.c_void_p(Dd.data_ptr()), C.c_void_p(Ee.data_ptr()), C.c_void_p(tau.data_ptr()))
        assert rc == 0, rc
    t_t, _ = timed(sytrd)
    t_e, _ = timed(lambda: torch.linalg.eigh(A))
    print(f"r {r}: syevd {t_e:.2f} ms | sytrd alone {t_t:.2f} ms | stedc alone {t_s:.2f} ms (sorted ascending: {bool((np.diff(wn) >= 0).all())}; |dw| {err_w:.1e} lmax, residual {res:.1e}, "
          f"Z^T Z - I {orth:.1e}, info {int(info)})", flush=True)
