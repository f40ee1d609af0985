"""ctypes binding of libvmcpde.so (include/vmcpde.h).  No torch types cross the boundary: only raw device
pointers (``tensor.data_ptr()``), sizes and the current HIP stream handle.

There is NO CPU fallback: if the library is missing this module raises at first use, and every entry point
raises ``VmcpdeError`` on a non-zero return code.
"""
from __future__ import annotations

import ctypes as C
import re
from pathlib import Path

import torch

_HERE = Path(__file__).resolve().parent
import os as _os

LIB_PATH = Path(_os.environ["VMCPDE_LIB"]) if _os.environ.get("VMCPDE_LIB") else _HERE / "_lib" / "libvmcpde.so"   # override: A/B builds
HEADER_PATH = _HERE.parent / "include" / "vmcpde.h"

MAX_DIM = 16
MAX_DEPTH = 16
MAX_HIDDEN_LAYERS = 3
MAX_DIRS = 136
GRAM_TILE = 128

COUPLING = {"no_add": 0, "different_add": 1, "jac_eq_1": 2, "plain": 3}
LATENT = {"Gauss": 0, "Student_t": 1}
EQ_KIND = {"diffusion": 0, "diffusion_drift": 1, "diffusion_anisotropic": 2, "advection_paper": 3,
           "advection_hamiltonian": 4, "advection_hamiltonian_wDiss": 5}


class VmcpdeError(RuntimeError):
    pass


class ModelDesc(C.Structure):
    _fields_ = [("dim", C.c_int32), ("depth", C.c_int32), ("n_hidden", C.c_int32),
                ("hidden", C.c_int32 * MAX_HIDDEN_LAYERS), ("coupling", C.c_int32), ("latent", C.c_int32),
                ("n_up", C.c_int32), ("inds_up", C.POINTER(C.c_int32)), ("inds_down", C.POINTER(C.c_int32)),
                ("offset", C.POINTER(C.c_double)), ("global_change", C.c_int32)]


class EqConsts(C.Structure):
    _fields_ = [("kind", C.c_int32), ("coupled", C.c_int32), ("D", C.c_double), ("mu", C.c_double), ("m", C.c_double), ("omega", C.c_double),
                ("lam", C.c_double), ("T", C.c_double), ("gamma", C.c_double)]


_vp, _i32, _i64, _u64, _f64, _int = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double, C.c_int

_PROTOS = {
    "vmcpde_last_error": (C.c_char_p, []),
    "vmcpde_abi_version": (_int, []),
    "vmcpde_create": (_int, [C.POINTER(ModelDesc), C.POINTER(_vp)]),
    "vmcpde_destroy": (_int, [_vp]),
    "vmcpde_num_parameters": (_int, [_vp]),
    "vmcpde_param_offset": (_int, [_vp, _int, _int, _int, _int]),
    "vmcpde_flow_eval_config": (_int, [_vp, _int, C.POINTER(_int), C.POINTER(_int)]),
    "vmcpde_sample_latent": (_int, [_vp, _vp, _u64, _u64, _i64, _i64, _vp, _vp]),
    "vmcpde_latent_from_normals": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "vmcpde_mcmc": (_int, [_u64, _u64, _i32, _i32, _f64, C.POINTER(_f64), _vp, _vp, _vp, _vp]),
    "vmcpde_mcmc_from_uniforms": (_int, [_vp, _vp, _vp, _i32, _i32, _f64, C.POINTER(_f64), _vp, _vp, _vp, _vp]),
    "vmcpde_flow_sample": (_int, [_vp, _vp, _vp, _i64, _int, _vp, _vp, _vp]),
    "vmcpde_flow_eval_workspace_bytes": (_i64, [_vp, _i64]),
    "vmcpde_flow_eval": (_int, [_vp, _vp, _vp, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _i64, _vp]),
    "vmcpde_flow_eval_from_latent": (_int, [_vp, _vp, _vp, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "vmcpde_rhs": (_int, [C.POINTER(EqConsts), _i32, _vp, _vp, _vp, _i32, _vp, _f64, _i64, _vp, _vp]),
    "vmcpde_eloc_stats": (_int, [_vp, _i64, _vp, _vp]),
    "vmcpde_row_moments": (_int, [_vp, _i64, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vmcpde_gram_packed_doubles": (_i64, [_i32]),
    "vmcpde_gram_workspace_bytes": (_i64, [_i32, _i64]),
    "vmcpde_gram": (_int, [_vp, _i64, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _i64, _vp]),
    "vmcpde_gram_rows": (_int, [_vp, _i64, _i32, _i64, _i32, _i32, _vp, _vp, _i64, _vp]),
    "vmcpde_gram_i8_workspace_bytes": (_i64, [_i32, _i64, _i32, _i32]),
    "vmcpde_gram_i8": (_int, [_vp, _i64, _i32, _i32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "vmcpde_gram_finalize": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _f64, _f64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    "vmcpde_tdvp_prepare_workspace_bytes": (_i64, []),
    "vmcpde_tdvp_prepare": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "vmcpde_tdvp_moments": (_int, [_vp, _i32, _i32, _vp, _vp, _vp, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _vp, _f64, _vp]),
    "vmcpde_tdvp_diagnostics": (_int, [_vp, _vp, _vp, _vp, _i32, _f64, _vp, _vp, _vp, _vp, _vp]),
    "vmcpde_observables_workspace_bytes": (_i64, []),
    "vmcpde_observables": (_int, [_vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _i64, _vp]),
    "vmcpde_allreduce": (_int, [_vp, _i64, _vp, _vp]),
    "vmcpde_comm_unique_id": (_int, [_vp]),
    "vmcpde_comm_init_rank": (_int, [C.POINTER(_vp), _i32, _vp, _i32]),
    "vmcpde_comm_destroy": (_int, [_vp]),
    "vmcpde_comm_info": (_int, [_vp, C.POINTER(_i32)]),
    "vmcpde_device_bus_id": (_int, [C.c_char_p, _i32]),
    "vmcpde_bench_mfma_f64": (_int, [_i32, _i32, _vp, _vp]),
    "vmcpde_chol_workspace_bytes": (_i64, [_i32]),
    "vmcpde_cholesky": (_int, [_vp, _i64, _i32, _vp, _vp, _i64, _vp]),
    "vmcpde_chol_solve": (_int, [_vp, _i64, _i32, _vp, _vp, _vp, _i64, _vp]),
    "vmcpde_chol_backsolve": (_int, [_vp, _i64, _i32, _vp, _vp, _vp, _i64, _vp]),
    "vmcpde_gemv": (_int, [_vp, _i64, _i32, _i32, _int, _vp, _vp, _vp]),
    "vmcpde_pinv_coeffs": (_int, [_vp, _vp, _vp, _i32, _f64, _f64, _vp, _vp]),
    "vmcpde_pivoted_cholesky_workspace_bytes": (_i64, [_i32]),
    "vmcpde_pivoted_cholesky": (_int, [_vp, _i64, _i32, _f64, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "vmcpde_pivoted_cholesky_blocked_workspace_bytes": (_i64, [_i32]),
    "vmcpde_pivoted_cholesky_blocked": (_int, [_vp, _i64, _i32, _f64, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "vmcpde_lowrank_candidates_byte_offset": (_i64, [_i32]),
    "vmcpde_lowrank_hint": (_int, [_vp, _vp, _i32, _vp, _i64, _vp]),
    "vmcpde_jt_matvec_workspace_bytes": (_i64, [_i32, _i64]),
    "vmcpde_jt_matvec": (_int, [_vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp, _i64, _vp]),
    "vmcpde_lowrank_columns": (_int, [_vp, _i32, _i32, _vp, _i64, _vp]),
    "vmcpde_lowrank_skip_flag_byte_offset": (_i64, []),
    "vmcpde_lowrank_begin": (_int, [_vp, _i32, _vp, _vp, _i64, _vp]),
    "vmcpde_lowrank_step": (_int, [_vp, _i32, _f64, _i32, _vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "vmcpde_gram_columns_workspace_bytes": (_i64, [_i32, _i64]),
    "vmcpde_gram_columns": (_int, [_vp, _i64, _i32, _i64, _vp, _vp, _f64, _vp, _i64, _vp, _vp, _i64, _vp]),
    "vmcpde_row_moments_sq": (_int, [_vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vmcpde_householder_rows": (_int, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    "vmcpde_woodbury_project": (_int, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vmcpde_woodbury_combine": (_int, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vmcpde_woodbury_prepare": (_int, [_vp, _f64, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _i32, _vp]),
    "vmcpde_woodbury_gram_workspace_bytes": (_i64, [_i64, _i32]),
    "vmcpde_woodbury_gram": (_int, [_vp, _i64, _i32, _vp, _vp, _i64, _vp, _i64, _vp]),
    "vmcpde_gather_words": (_int, [C.POINTER(_vp), C.POINTER(_i32), _i32, _vp, _vp]),
    "vmcpde_frozen_gather": (_int, [_vp, _i64, _i32, _vp, _i32, _i32, _vp, _i64, _vp]),
    "vmcpde_cholesky_tall": (_int, [_vp, _i64, _i32, _i32, _vp, _vp, _i64, _vp]),
    "vmcpde_frozen_finish": (_int, [_vp, _i64, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _vp]),
    "vmcpde_frozen_check": (_int, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "vmcpde_jacobi_shape": (_int, [_i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "vmcpde_jacobi_columns": (_int, [_vp, _i32, _i32, _vp, _vp]),
    "vmcpde_sytrd_small_max_n": (_i32, []),
    "vmcpde_sytrd_small": (_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _vp]),
    "vmcpde_apply_reflectors": (_int, [_vp, _i64, _vp, _i32, _vp, _i64, _i32, _vp]),
    "vmcpde_sytrd_medium_max_n": (_i32, []),
    "vmcpde_sytrd_medium_workspace_bytes": (_i64, [_i32]),
    "vmcpde_sytrd_medium": (_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp]),
}

_lib = None


def header_symbols():
    """Every entry point include/vmcpde.h declares (used by the CPU test that the library exports them all)."""
    text = HEADER_PATH.read_text()
    return sorted(set(re.findall(r"\b(vmcpde_[a-z0-9_]+)\s*\(", text)))


def lib():
    """Load libvmcpde.so (built by ``python -m vmc_pde_amd.build``).  Fails loudly when it is absent."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise VmcpdeError(f"{LIB_PATH} not found: the HIP library is the only compute path of vmc_pde_amd "
                              f"(no CPU fallback). Build it with `python -m vmc_pde_amd.build`.")
        l = C.CDLL(str(LIB_PATH))
        for name, (res, args) in _PROTOS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().vmcpde_last_error().decode(errors="replace")
        raise VmcpdeError(f"{what or 'libvmcpde'} failed (rc={rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Tensors must be fp64/intN contiguous CUDA(HIP) memory."""
    if t is None:
        return None
    assert t.is_cuda, "libvmcpde operates on device (HBM) buffers only"
    return C.c_void_p(t.data_ptr())


def stream():
    """Handle of torch's current HIP stream: every kernel is launched on it."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def call(name: str, *args):
    check(getattr(lib(), name)(*args), name)
