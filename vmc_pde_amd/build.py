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
SOURCES = ["model.hip", "flow.hip", "sampler.hip", "gram.hip", "moments.hip", "solve.hip", "comm.hip"]
HEADERS = [CSRC / "common.hpp", CSRC / "mfma_tile.hpp", PKG.parent / "include" / "vmcpde.h"]
ARCH = "gfx950"


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


def build(force: bool = False, verbose: bool = False) -> Path:
    LIBDIR.mkdir(exist_ok=True)
    objs = []
    flags = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fgpu-rdc" if False else "-fno-gpu-rdc",
             "-Wall", "-Wno-unused-function", "-ffp-contract=on"]
    for src in SOURCES:
        s = CSRC / src
        o = LIBDIR / (src.replace(".hip", ".o"))
        if force or _stale(o, [s] + HEADERS):
            cmd = [_hipcc(), *flags, "-c", str(s), "-o", str(o)]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        objs.append(o)
    if force or _stale(LIB, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB), *map(str, objs), "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv)
    print(p)
