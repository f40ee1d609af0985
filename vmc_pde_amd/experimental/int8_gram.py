"""EXPERIMENT (opt-in, never the headline): the S-build on the INT8 matrix pipe by error-free slicing of the Jacobian.

csrc/gram_i8.hip: six balanced 8-bit digits per fp64 entry (four for an fp32 Jacobian), the digit-pair products summed exactly in
int32 by v_mfma_i32_32x32x32_i8 and combined in fp64 -- 1.4x the fp64 MFMA syrk at the headline size, S to ~1e-13 |S|_F of it
(DESIGN.md section 4).  north_star prescribes the fp64 MFMA syrk; this variant is reported beside it.
"""
from __future__ import annotations

import torch

from .. import _native as nat
from .. import global_defs
from .. import mpi_wrapper as mpi


class Int8GramMixin:
    """Overrides `_gram_launch` and `_lowrank_tol` of `tdvp.TDVP`; state: the asynchronously read heavy-tail figure of the previous
    int8 call (`_i8_host`, `_i8_event`), the switch-off flag `_i8_off`, the digit count of the last call `_i8_digits`."""

    def _init_int8(self):
        self._i8_host = self._i8_event = None
        self._i8_off = False
        self._i8_digits = 6

    def _int8_tail_limit(self):
        """Largest row max / row rms figure the int8 Gram is trusted with.  Measured (tools/probe_gram_i8.py, 65 536 x 4096): the error of the
        six-digit form is 1.2e-13 of |S|_F at a figure of 8 (Gaussian rows) and 1.9e-12 at 68 (cubed normals), i.e. ~1.5e-14 x figure^1.3;
        the four-digit form used for an fp32 Jacobian (31 bits) is 5e4 times that.  The estimate must stay a factor ten below what
        regularises the solve: the spectral cut-off svdTol, or for the Cholesky the larger of svdTol (absolute floor) and diagonalShift.
        (Conservative: rows dominated by a few samples -- the model's own Jacobians reach figures of 150-330 -- are summed far more
        accurately than this, tools/diag/i8_real_jacobian.py; the figure never exceeds ~2 sqrt(N).)"""
        reg = max(float(self.svdTol), float(self.diagonalShift) if self._solver() == "cholesky" else 0.0, 1e-16)
        coeff = 1.5e-14 * (5e4 if self._i8_digits == 4 else 1.0)
        return (0.1 * reg / coeff) ** (1.0 / 1.3)

    def _int8_still_ok(self):
        """The per-row scale of the sliced Gram makes its error grow like (row max / row rms)^2: the previous int8 call's figure is
        read here (asynchronous copy, as for the uncentred Gram's mean^2 / var check); above the limit the engine warns once and
        stays on the fp64 kernels."""
        if self._i8_event is not None and not self._i8_off:
            self._i8_event.synchronize()
            ratio, limit = float(self._i8_host[0]), self._int8_tail_limit()
            if ratio > limit:
                import warnings
                warnings.warn(f"int8Gram: a Jacobian row has max / rms ~ {ratio:.3g} > {limit:.3g} (the bound for svdTol = {self.svdTol:g}, "
                              f"diagonalShift = {self.diagonalShift:g}); the sliced Gram would lose accuracy there -- using the fp64 MFMA "
                              f"kernel from now on", RuntimeWarning)
                self._i8_off = True
        return not self._i8_off


    def _gram_launch(self, b, JT, ldj, f32, P, n, c, w, out):
        if not (self.int8Gram and f32 in (0, 1) and self._int8_still_ok()):
            return super()._gram_launch(b, JT, ldj, f32, P, n, c, w, out)
        # six digits for an fp64 Jacobian (47 bits); four for an fp32 one (31 bits: every stored value to 2^-31 of its row maximum)
        ns = self._i8_digits = 6 if f32 == 0 else 4
        need8 = nat.lib().vmcpde_gram_i8_workspace_bytes(P, n, ns, 8)
        if b.get("gram_i8_ws") is None or b["gram_i8_ws"].numel() * 8 < need8:
            b["gram_i8_ws"] = torch.empty(need8 // 8 + 1, dtype=global_defs.tReal, device=JT.device)
            b["i8_stats"] = torch.zeros(1, dtype=global_defs.tReal, device=JT.device)
        nat.call("vmcpde_gram_i8", nat.ptr(JT), ldj, f32, P, n, ns, 8, nat.ptr(c), nat.ptr(w), nat.ptr(out), nat.ptr(b["i8_stats"]),
                 nat.ptr(b["gram_i8_ws"]), b["gram_i8_ws"].numel() * 8, nat.stream())
        # heavy-tail figure of this call (largest row max / row rms) for the host, without synchronising: read before the next call.
        # Sharded samples: the LARGEST figure of any rank decides for all of them (every rank must build the next S the same way and
        # give the replicated low-rank routes the same tolerance)
        if mpi.get_comm_size() > 1:
            mpi.allreduce_max_(b["i8_stats"])
        if self._i8_host is None:
            self._i8_host = torch.empty(1, dtype=global_defs.tReal).pin_memory()
            self._i8_event = torch.cuda.Event()
        self._i8_host.copy_(b["i8_stats"], non_blocking=True)
        self._i8_event.record()

    def _lowrank_tol(self):
        """The six-digit int8 S-build is 1.2e-13 |S|_F away from the fp64 one: pivoting stops at 1e-12.  Four int8 digits (fp32 Jacobian:
        6e-9) bury everything below ~1e-8 max diag in noise -- such an S has no usable numerical rank (measured: a plan made on the first,
        fp64-built S failed the residual guard on the first int8-built one), so that variant keeps the dense paths."""
        tol = super()._lowrank_tol()
        if tol is None or not self.int8Gram or self._i8_off:
            return tol
        return None if self._i8_digits == 4 else max(tol, 1e-12)
