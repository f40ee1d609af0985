"""CPU checks of the drop-in boundary: the C-ABI library builds for gfx950 without a GPU, loads, and exports every
symbol include/vmcpde.h declares (no compute calls here); the product path fails loudly when the library is missing."""
import ctypes
import subprocess

import pytest


def test_library_exports_every_header_symbol(lib_built):
    from vmc_pde_amd import _native as nat
    syms = nat.header_symbols()
    assert len(syms) >= 25
    lib = ctypes.CDLL(str(lib_built))
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert set(nat._PROTOS) == set(syms)            # every entry point has a ctypes prototype and vice versa
    assert nat.lib().vmcpde_abi_version() == 7
    nm = subprocess.run(["nm", "-D", "--defined-only", str(lib_built)], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in nm.splitlines() if " T " in l}
    assert set(syms) <= exported


def test_argument_validation_needs_no_gpu(lib_built):
    # host-side checks run before any HIP call: error codes + messages instead of crashes
    from vmc_pde_amd import _native as nat
    L = nat.lib()
    assert L.vmcpde_gram_packed_doubles(4213) == (33 * 34 // 2) * 128 * 128
    assert L.vmcpde_gram_packed_doubles(128) == 128 * 128
    ws = L.vmcpde_gram_workspace_bytes(4213, 65536)
    slab, counters = (33 * 34 // 2) * 128 * 128 * 8, ((33 * 34 // 2) * 4 + 255) // 256 * 256     # (round 6: one counter per tile behind the slabs)
    assert (ws - counters) % slab == 0 and 8 <= (ws - counters) // slab <= 12                        # split-K factor
    assert L.vmcpde_chol_workspace_bytes(129) == 2 * 128 * 128 * 8 + 256      # two inverse blocks + the counters of the chained launches
    assert L.vmcpde_num_parameters(None) == -1
    rc = L.vmcpde_gram(None, 0, 0, 0, 0, None, None, None, None, 0, None)
    assert rc == -1 and b"null" in L.vmcpde_last_error()
    with pytest.raises(nat.VmcpdeError):
        nat.call("vmcpde_cholesky", None, 0, 0, None, None, 0, None)


def test_no_cpu_fallback_when_library_is_missing(monkeypatch, tmp_path):
    from vmc_pde_amd import _native as nat
    monkeypatch.setattr(nat, "_lib", None)
    monkeypatch.setattr(nat, "LIB_PATH", tmp_path / "libvmcpde.so")
    with pytest.raises(nat.VmcpdeError, match="no CPU fallback"):
        nat.lib()


def test_product_never_imports_the_oracle():
    import pathlib
    root = pathlib.Path(__file__).resolve().parents[1] / "vmc_pde_amd"
    for f in root.rglob("*.py"):
        assert "oracle" not in f.read_text(), f


def test_rocsolver_binding_resolves_without_a_gpu():
    """vmc_pde_amd/rocsolver.py binds ONE library routine -- rocSOLVER's tridiagonal divide & conquer, which runs between the library's own
    tridiagonalisation and back-transformation kernels (csrc/tridiag.hip) -- from the libraries the running torch ships: the symbol must
    resolve here (no compute call without a GPU); if it ever does not, `available()` says so and the spectral core falls back to syevd."""
    from vmc_pde_amd import rocsolver
    assert rocsolver.available() in (True, False)
    if rocsolver.available():
        blas, solver = rocsolver._load()
        assert hasattr(solver, "rocsolver_dstedc") and hasattr(blas, "rocblas_create_handle") and hasattr(blas, "rocblas_set_stream")
