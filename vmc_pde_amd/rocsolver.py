"""rocSOLVER's tridiagonal divide-and-conquer eigensolver (stedc) through ctypes.

The r x r core of the rank-revealing spectral solve (reference: vmc_fluids/tdvp.py:57-64, `eigh`) is tridiagonalised by the library's own
kernel (csrc/tridiag.hip: the dependent chain that makes a library syevd slow at this size) and back-transformed by another; what runs in
between -- the eigen-decomposition of a tridiagonal matrix -- is a plain library call, which torch does not expose on its own.  The
libraries are the ones the running torch has already loaded (no second copy).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_EVECT_TRIDIAGONAL = 212      # rocblas_evect_tridiagonal
_libs = None
_handles = {}


class RocsolverUnavailable(RuntimeError):
    pass


def _load():
    global _libs
    if _libs is None:
        libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
        try:
            blas = C.CDLL(os.path.join(libdir, "librocblas.so"))
            solver = C.CDLL(os.path.join(libdir, "librocsolver.so"))
            solver.rocsolver_dstedc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
            solver.rocsolver_dstedc.restype = C.c_int
            blas.rocblas_create_handle.argtypes = [C.POINTER(C.c_void_p)]
            blas.rocblas_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        except (OSError, AttributeError) as exc:
            _libs = exc
        else:
            _libs = (blas, solver)
    if isinstance(_libs, Exception):
        raise RocsolverUnavailable(f"rocSOLVER / rocBLAS of the running torch could not be bound: {_libs}")
    return _libs


def available() -> bool:
    try:
        _load()
        return True
    except RocsolverUnavailable:
        return False


def _handle(device):
    key = torch.device(device).index or 0
    h = _handles.get(key)
    if h is None:
        blas, _ = _load()
        h = C.c_void_p()
        with torch.cuda.device(key):
            rc = blas.rocblas_create_handle(C.byref(h))
        if rc != 0:
            raise RocsolverUnavailable(f"rocblas_create_handle failed ({rc})")
        _handles[key] = h
    return h


def stedc(D: torch.Tensor, E: torch.Tensor, Z: torch.Tensor, info: torch.Tensor):
    """Eigenpairs of the symmetric tridiagonal matrix (diagonal D, sub-diagonal E[:n-1]) on torch's current stream: D <- eigenvalues
    (ascending), row m of the n x n row-major Z <- eigenvector m (the column-major matrix LAPACK's dstedc returns), info[0] <- 0 or the
    number of unconverged blocks.  E is destroyed.  Nothing synchronises."""
    n = D.numel()
    assert D.is_cuda and D.dtype == torch.float64 and E.numel() >= n - 1 and Z.shape == (n, n) and Z.is_contiguous() and info.dtype == torch.int32
    blas, solver = _load()
    h = _handle(D.device)
    with torch.cuda.device(D.device):
        rc = blas.rocblas_set_stream(h, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc == 0:
            rc = solver.rocsolver_dstedc(h, _EVECT_TRIDIAGONAL, n, C.c_void_p(D.data_ptr()), C.c_void_p(E.data_ptr()), C.c_void_p(Z.data_ptr()), n,
                                         C.c_void_p(info.data_ptr()))
    if rc != 0:
        raise RuntimeError(f"rocsolver_dstedc failed (status {rc})")
