"""Build libvmcpde.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m vmc_pde_amd.build [--force] [--verbose]

The shared library lands in vmc_pde_amd/_lib/ (git-ignored, but it travels with the gpurun snapshot).
"""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "_lib"
LIB = LIBDIR / "libvmcpde.so"
SOURCES = ["model.hip", "flow.hip", "sampler.hip", "gram.hip", "gram_i8.hip", "moments.hip", "solve.hip", "eig.hip", "jacobi.hip", "tridiag.hip", "comm.hip"]
HEADERS = [CSRC / "common.hpp", CSRC / "mfma_tile.hpp", CSRC / "gram_order.hpp", CSRC / "wave_ops.hpp", CSRC / "flow_kernels.inc", PKG.parent / "include" / "vmcpde.h"]
ARCH = "gfx950"
# per-file flags.  solve.hip: keep MFMA accumulators in VGPRs -- the diagonal-block sweep edits them with VALU instructions between
# MFMAs, and with the default AGPR form every step pays 40 v_accvgpr moves (measured: the whole factorisation 0.1 ms slower)
FILE_FLAGS = {"solve.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    return "hipcc"


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False, variant: str = "", extra_flags=()) -> Path:
    """The product library, or -- `variant` + `extra_flags` -- an A/B build (e.g. -DVMC_DIAG_PHASE_KNOB) that lands in
    _lib/libvmcpde_<variant>.so and is selected at run time with VMCPDE_LIB."""
    LIBDIR.mkdir(exist_ok=True)
    objdir = LIBDIR / variant if variant else LIBDIR
    objdir.mkdir(exist_ok=True)
    lib = LIBDIR / f"libvmcpde_{variant}.so" if variant else LIB
    objs = []
    flags = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc",
             "-Wall", "-Wno-unused-function", "-ffp-contract=on", *extra_flags]
    for src in SOURCES:
        s = CSRC / src
        o = objdir / (src.replace(".hip", ".o"))
        if force or _stale(o, [s] + HEADERS):
            cmd = [_hipcc(), *flags, *FILE_FLAGS.get(src, []), "-c", str(s), "-o", str(o)]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        objs.append(o)
    if force or _stale(lib, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(lib), *map(str, objs), "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return lib


if __name__ == "__main__":
    # python -m vmc_pde_amd.build [--force] [-v] [--variant NAME -DFLAG ...]
    variant = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else ""
    p = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv, variant=variant,
              extra_flags=[a for a in sys.argv[1:] if a.startswith("-D")])
    print(p)
