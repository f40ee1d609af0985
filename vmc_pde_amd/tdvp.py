"""TDVP / stochastic-reconfiguration engine (reference: vmc_fluids/tdvp.py).

`TDVP(useSNR, snrTol, svdTol, diagonalShift, diagonalizeOnDevice)` and `__call__(netParameters, t, psi, evolutionEq,
**rhsArgs) -> (update, info)` keep the reference's call surface (tdvp.py:20-26, 96-164) and publish the same
attributes (S0, F0, SExp, S, ev, V, VtF, snr, rhoVar, ElocMean, ElocMeanAbs, ElocVar, solverResidual, tdvp_error).

What runs underneath (all on the GPU, all fp64):
  * Eloc statistics, row moments, ONE fp64-MFMA syrk with fused centring (csrc/gram.hip) instead of the reference's
    four N x P temporaries and two full GEMMs (tdvp.py:40-47);
  * multi-GPU: every rank processes its shard of the samples and ONE packed buffer
    [ Gram tiles | sum O | sum E~O | sum E, sum E^2, sum |E| | n ] is all-reduced (RCCL) per step; the solve is replicated;
  * solve: `solver="cholesky"` -- in-place blocked Cholesky of S + diagonalShift*diag(S) + svdTol*max(diag S)*I on the MFMA tile engine
    (csrc/solve.hip); `solver="eigh"` -- the reference's spectral pseudo-inverse (tdvp.py:57-94), with eigh from
    LAPACK on the host (diagonalizeOnDevice=False, as tdvp.py:61-64) or rocSOLVER on the device.
  * `SExp`, `rhoVar`, `snr` are evaluated lazily from the retained Jacobian (weighted Gram), since only AdaptiveHeun
    (stepper.py:71) and useSNR / the driver's logging (main.py:187) read them.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _native as nat
from . import global_defs
from . import mpi_wrapper as mpi


F32_BLOCK_MIN_SHIFT = 1e-4   # smallest diagonalShift the fp32-block Gram is allowed with (see TDVP._gram)
_BORDER_PIVOT = 1e200    # corner of the bordered matrix: any value above F^T A^-1 F keeps the last pivot positive


def _long_k_product(A, B):
    """A @ B^T for row-major A (m x K) and B (n x K) with a long contraction and a small result: as ONE library GEMM such a shape runs on
    a handful of workgroups (256 x 256 x 4213: 0.23 ms), as a batched GEMM over chunks of the contraction with the chunk sums added
    up it fills the chip (0.03 ms).  Rows may be strided (views into padded buffers)."""
    K = A.shape[1]
    if K < 1024:
        return A @ B.T
    if K % 64:                                         # (a zero pad of the contraction: two small copies)
        pad = 64 - K % 64
        A, B = torch.nn.functional.pad(A, (0, pad)), torch.nn.functional.pad(B, (0, pad))
        K += pad
    ck = 128 if K % 128 == 0 else 64
    A3 = A.view(A.shape[0], K // ck, ck).permute(1, 0, 2)
    B3 = B.view(B.shape[0], K // ck, ck).permute(1, 2, 0)
    return torch.bmm(A3, B3).sum(0)


def _lapack_threads(P):
    """Thread cap for the host eigh: OpenBLAS' default (one thread per hardware thread, 64+) makes a 411 x 411 eigh take 94 ms
    instead of 9 ms with one thread, and a 4213 x 4213 one 5.3 s instead of 1.3 s with 16 (measured on the GPU box's host)."""
    global _THREADPOOLS
    limit = 1 if P < 600 else (8 if P < 2000 else 16)
    try:
        if _THREADPOOLS is None:     # (threadpool_limits() re-scans the loaded libraries on every call: 0.45 ms; the controller is made once)
            from threadpoolctl import ThreadpoolController
            _THREADPOOLS = ThreadpoolController()
        return _THREADPOOLS.limit(limits=limit)
    except ImportError:          # no cap available: plain LAPACK call
        import contextlib
        return contextlib.nullcontext()


_THREADPOOLS = None


DEVICE_EIGH_MIN_P = 640     # diagonalizeOnDevice=None: rocSOLVER's syevd overtakes thread-capped host LAPACK between P = 411 and 1024
LOWRANK_MIN_P = 1024        # spectralEngine="auto": the rank-revealing decomposition from this many parameters on (below, syevd takes < 25 ms)
LOWRANK_RANK_CAP = 768      # ... while the numerical rank stays below this and below P / LOWRANK_RANK_DIVISOR: its chains are r dependent
LOWRANK_RANK_DIVISOR = 6    #   launches of O(P r) work each, and the r x r core goes to syevd
LOWRANK_TOL = 1e-14         # pivoting stops when the largest remaining diagonal entry is <= this x max diag(S).  The downdated diagonal of a row
                            #   carries round-off of ~5 eps S_ii, and a pivot p turns noise n into a column of size n / sqrt(p): pivots must
                            #   stay an order of magnitude above 1e-15 max diag (with 1e-16 the last columns added 4e-13 max diag to L L^T on
                            #   a 700 x 700 test matrix; LAPACK's dpstrf stops at P eps max diag = 9e-13 here).  max diag <= lambda_max, so
                            #   everything dropped lies below the reference's hard cut-off at 1e-14 lambda_max (tdvp.py:82)
LOWRANK_TRACE_BOUND = 1e-11 # accept only if the trace of the remainder (an upper bound of its norm; <= (P - r) LOWRANK_TOL by construction: 1.1e-12
                            #   at P = 16 181) is below this x max diag(S)
LOWRANK_CHOL_MIN_P = 4096   # choleskyEngine="auto": the low-rank solve from this many parameters on.  P = 4213: 1.3 ms against 2.1 ms for the blocked
                            #   factorisation of the full matrix (which scales as P^3: 0.8 ms at 3072) -- in the static mode, i.e. without a host
                            #   read in the step; with one (planning mode) 2.5 ms.  P = 16 181: 7 against 38 ms
LOWRANK_SOLVE_RHO = 1e-4    # low-rank Cholesky solve: largest trace(R) / min D accepted (the contraction factor of its iterative refinement)
LOWRANK_RETRY_AFTER = 50    # "auto": calls on the dense path after one whose rank exceeded the cap, before the next attempt


def _round_up(v, m):
    return (v + m - 1) // m * m


class StaticPlanOutdated(FloatingPointError):
    """The numerical rank of S outgrew the static plan of the low-rank solve within one call, so that call's update failed the residual
    guard.  The plan has been dropped: repeating the step (the steppers do, once: `retry_step`) runs a planning call."""
    retry_step = True


class _VecTimesGram:
    """`v @ S` for a GramOperator S, pending: `(v @ S) @ v` -- how the reference's norm function spells the quadratic form
    (main.py:24-26) -- is then ONE pass over the Jacobian; anything else turns it into the vector S v."""

    def __init__(self, op, v):
        self._op, self._v = op, v

    def __matmul__(self, w):
        if torch.is_tensor(w) and w.dim() == 1 and (w is self._v or (w.data_ptr() == self._v.data_ptr() and w.shape == self._v.shape)):
            return self._op.quadratic_form(self._v)
        return self.vector() @ w

    def vector(self):
        return self._op.matvec(self._v)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        args = tuple(a.vector() if isinstance(a, _VecTimesGram) else a for a in args)
        return func(*args, **(kwargs or {}))

    def __getattr__(self, name):
        return getattr(self.vector(), name)


class GramOperator:
    """SExp = <(logp O~)(logp O~)^T> (tdvp.py:47) as the adaptive stepper uses it -- the matrix of a norm, `v @ S @ v` (stepper.py:71,
    main.py:24-26) -- without forming it: v^T SExp v = (1/N) sum_n (logp_n (J_n - m).v)^2 is one pass over the retained Jacobian
    (`vmcpde_jt_matvec`, first pass: 0.4 ms at the headline size) where the P x P matrix is a second Gram (17 ms).  `v @ S @ v`,
    `S @ v` and `torch.dot(v, S @ v)` take the passes; every other use (indexing, arithmetic, `.cpu()`, ...) gets the dense matrix of
    `TDVP.SExp`, formed on that first use."""

    def __init__(self, tdvp):
        self._t = tdvp

    def dense(self):
        return self._t.SExp

    def quadratic_form(self, v):
        return self._t._sexp_form(v)

    def matvec(self, v):
        return self._t._sexp_matvec(v)

    def __matmul__(self, v):
        if torch.is_tensor(v) and v.dim() == 1:
            return self.matvec(v)
        return self.dense() @ v

    def __rmatmul__(self, v):
        if torch.is_tensor(v) and v.dim() == 1:
            return _VecTimesGram(self, v)
        return v @ self.dense()

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", "")
        if name in ("matmul", "__matmul__", "mv") and len(args) == 2 and not kwargs:
            a, b = args
            if isinstance(b, GramOperator) and torch.is_tensor(a):
                return b.__rmatmul__(a)
            if isinstance(a, GramOperator) and torch.is_tensor(b):
                return a.__matmul__(b)
        if name == "__rmatmul__" and len(args) == 2 and not kwargs and isinstance(args[0], GramOperator) and torch.is_tensor(args[1]):
            return args[0].__rmatmul__(args[1])
        args = tuple(a.dense() if isinstance(a, GramOperator) else a for a in args)
        return func(*args, **kwargs)

    def __getattr__(self, name):
        return getattr(self.dense(), name)

    def __getitem__(self, idx):
        return self.dense()[idx]


@dataclass
class TDVP:
    useSNR: bool = False
    snrTol: float = 2e0
    svdTol: float = 1e-11
    diagonalShift: float = 0
    diagonalizeOnDevice: object = None   # tdvp.py:24,59-64.  False: host LAPACK (the reference's default), True: rocSOLVER on the device,
                                  # None (default here): by size -- host below DEVICE_EIGH_MIN_P parameters, device from there on
                                  # (P = 411: 5.2 ms host / 8.2 ms device; P = 1024: 57 / 22 ms; P = 4213: 864 / 147 ms;
                                  # tools/microbench/eigh_variants.py).  Same decomposition either way; if rocSOLVER does not
                                  # converge on a singular S the call falls back to the host with a warning.
    solver: str = "auto"          # "eigh": the reference's spectral pseudo-inverse (tdvp.py:57-94) -- what "auto" selects, so that the
                                  #   reference's constructor arguments solve the reference's system and publish ev / V / snr;
                                  # "cholesky": explicit opt-in to the on-device regularised Cholesky (no eigenbasis, hence no SNR)
    spectralEngine: str = "auto"  # how the spectral path obtains (ev, V) on the device (tdvp.py:57-64).  "dense": rocSOLVER's syevd on the full
                                  #   matrix (147 ms at P = 4213).  "lowrank": the rank-revealing decomposition of csrc/eig.hip -- pivoted
                                  #   Cholesky of S, Householder QR of the factor, Rayleigh-Ritz on its range, V = [null space | Ritz
                                  #   vectors] -- O(P r^2) for numerical rank r (238 of 4213 at the headline size); the same (ev, V) to
                                  #   the backward error of a dense eigensolver, the ~P - r eigenvalues below 1e-16 of the largest
                                  #   reported as exact zeros.  "auto": lowrank from LOWRANK_MIN_P parameters on when diagonalShift
                                  #   leaves S rank deficient, dense whenever the rank exceeds the cap (then not retried for a while).
    choleskyEngine: str = "auto"  # how solver="cholesky" solves (S + diagonalShift diag(S) + floor I) u = F.  "dense": the blocked factorisation of
                                  #   the full matrix (csrc/solve.hip: 2.0 ms at P = 4213, 36 ms at 16 181).  "lowrank": S = L L^T + R by the
                                  #   pivoted Cholesky of csrc/eig.hip (r = 190 columns of 4213), the diagonal part D through the Woodbury
                                  #   identity -- a Cholesky of the r x r matrix I + L^T D^-1 L -- and one or two steps of iterative refinement
                                  #   against the TRUE matrix S + D, which remove what R and the identity's cancellation left (the solution
                                  #   of the same system to ~1e-14).  "auto": lowrank from LOWRANK_CHOL_MIN_P parameters on (where the O(P^3)
                                  #   factorisation costs more than the r dependent block steps) while the rank stays under the cap and
                                  #   trace(R) under 1e-4 min D; dense otherwise (and for the next LOWRANK_RETRY_AFTER calls).
    matrixFree: bool = False      # EXPERIMENT (opt-in, solver="cholesky", fp64 Jacobian): S is never assembled.  The pivoted low-rank
                                  # factor of S = <J^T J> needs only diag(S) and the columns it pivots on, which are formed from the
                                  # Jacobian: for the previous call's pivots AHEAD in one pass (pivotHints), 32 at a time on demand otherwise
                                  # (csrc/eig.hip::colpanel_kernel), and S0 u for the refinement and the diagnostics is J^T (J u) / N - m (m.u)
                                  # (vmcpde_jt_matvec): ~7 passes over J and a 224-column GEMM instead of N P^2 flop.
                                  # `S0` is then formed on first access; a full-rank S (no plan within the rank cap) raises.
    pivotHints: bool = True       # low-rank routes: the pivots of the previous call's factorisation are the preferred candidates of this
                                  # one (`vmcpde_lowrank_hint`; consecutive S of a trajectory are close).  Only the choice of candidates
                                  # changes -- acceptance, window and stopping criterion do not --, about 22 instead of 13 columns per block
                                  # step; a call's result then depends on the call before it at the level of the solve's accuracy (1e-10).
    residualBound: float = 1e-6   # Cholesky path: a solver residual above this (or a NaN) is treated as a failed factorisation
    centering: str = "auto"       # "exact": global means before the Gram (an extra pass over J, + one tiny all-reduce when sharded)
                                  # "lagged": previous call's global mean as shift -> one pass, a single all-reduce per step
                                  # "auto": exact on the first call, lagged afterwards (algebraically identical, see finalize)
    gramShift: str = "auto"       # "centred": the Gram kernel subtracts the shift while staging (as tdvp.py:42 centres first)
                                  # "uncentred": raw sum J J^T, mean outer product removed in finalize (no VALU work in the
                                  #   kernel's staging burst, which competes with the fp64 MFMAs for issue slots)
                                  # "auto": uncentred once the previous call showed max_p mean_p^2/var_p < 1e3, i.e. the
                                  #   cancellation costs < 3 of fp64's 16 digits; centred otherwise
    fuseMoments: bool = True      # uncentred path: sum O and sum O (E - cE) ride in the syrk as two appended Jacobian rows
    f32BlockGram: bool = False    # with an fp32 Jacobian: fp32 MFMA sums over blocks of <= 128 samples, fp64 across blocks (2x MFMA rate)
    int8Gram: bool = False        # EXPERIMENT (opt-in): the uncentred fp64 Gram on the INT8 matrix pipe by error-free slicing of the Jacobian
                                  # (csrc/gram_i8.hip: six balanced 8-bit digits per entry, 21 digit-pair products summed exactly in
                                  # int32, combined in fp64) -- 1.4x the fp64 MFMA syrk at C2, S to ~1e-13 |S|_F of it; also for the
                                  # centred and the weighted (SExp, rhoVar) sums; an fp32 Jacobian keeps the fp64-accumulating kernels
    frozenParameters: object = None   # flat parameter indices held fixed: their rows / columns of S and entries of F are zeroed, so
                                  # both solvers return a zero update for them (the pseudo-inverse drops the zero modes, tdvp.py:82;
                                  # the Cholesky pins or floors them).  Not a reference argument: the published diffusion runs used
                                  # a latent covariance I + A A^T (main.py:55-57) whose gradient vanishes at the initial A = 0, which
                                  # is this model with L, L_diag frozen.

    def __post_init__(self):
        if self.solver not in ("auto", "eigh", "cholesky"):
            raise ValueError(f"unknown solver {self.solver!r}")
        if self.spectralEngine not in ("auto", "dense", "lowrank"):
            raise ValueError(f"unknown spectralEngine {self.spectralEngine!r}")
        if self.choleskyEngine not in ("auto", "dense", "lowrank"):
            raise ValueError(f"unknown choleskyEngine {self.choleskyEngine!r}")
        self._chol_backoff = 0
        self._chol_static = None
        self._mf_plan = None
        self.choleskyInfo = None
        self._lowrank_backoff = 0
        self._lowrank_last_rank = 0
        self.spectralInfo = None
        if self.solver == "cholesky" and self.useSNR:
            # tdvp.py:87-89: the SNR soft cut-off lives in the eigenbasis of S; a Cholesky factor has none
            raise ValueError("useSNR=True needs the spectral solver (solver='eigh' or 'auto'); the Cholesky path has no eigenbasis")
        self._guard_host = self._guard_event = None
        self._guard_pending = self._guard_chol = False
        self._i8_host = self._i8_event = None
        self._i8_off = False
        self._i8_digits = 6
        self.cholInfo = None
        self._buf = {}
        self._c_prev = None
        self._lazy = None
        self._SExp = self._rhoVar = self._snr = self._S = None
        self._S0, self._S0_lazy = None, False
        self.ev = self.V = self.VtF = None
        self._ratio_host = self._ratio_event = None
        self._uncentred_ok = False
        self._floor = None

    # ---- buffers ------------------------------------------------------------------------------------------------
    def _buffers(self, P, device):
        b = self._buf.get((P, str(device)))
        if b is None:
            Ppad = _round_up(P, nat.GRAM_TILE)
            npk = nat.lib().vmcpde_gram_packed_doubles(P)
            pack = torch.zeros(npk + 2 * P + 4, dtype=global_defs.tReal, device=device)
            b = dict(Ppad=Ppad, npk=npk, pack=pack,
                     packed=pack[:npk], rowsum=pack[npk:npk + P], rowdot=pack[npk + P:npk + 2 * P],
                     sums=pack[npk + 2 * P:npk + 2 * P + 3], count=pack[npk + 2 * P + 3:npk + 2 * P + 4],
                     stats=torch.zeros(4, dtype=global_defs.tReal, device=device),
                     S=torch.zeros((Ppad, Ppad), dtype=global_defs.tReal, device=device),
                     F=torch.zeros(P, dtype=global_defs.tReal, device=device),
                     diagS=torch.zeros(P, dtype=global_defs.tReal, device=device),
                     prep_ws=torch.zeros(nat.lib().vmcpde_tdvp_prepare_workspace_bytes() // 8, dtype=global_defs.tReal, device=device),
                     gram_ws=None, A=None, chol_ws=None, info=torch.zeros(1, dtype=torch.int32, device=device))
            self._buf = {(P, str(device)): b}
        return b

    class _Section:
        """HIP events on the launch stream around one stage of a call, handed to `TDVP._event_hook(name, e0, e1)` when bench.py
        has set one (no cost otherwise)."""

        def __init__(self, owner, name):
            self.hook, self.name = getattr(owner, "_event_hook", None), name

        def __enter__(self):
            if self.hook is not None:
                self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                self.e0.record()
            return self

        def __exit__(self, *exc):
            if self.hook is not None and exc[0] is None:
                self.e1.record()
                self.hook(self.name, self.e0, self.e1)
            return False

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

    def _gram(self, b, JT, ldj, P, n, c, w, out):
        need = nat.lib().vmcpde_gram_workspace_bytes(P, n)
        if b["gram_ws"] is None or b["gram_ws"].numel() * 8 < need:
            b["gram_ws"] = torch.empty(max(need // 8, 1), dtype=global_defs.tReal, device=JT.device)
        f32 = 0 if JT.dtype != torch.float32 else (2 if (self.f32BlockGram and c is None and w is None and ldj % 4 == 0) else 1)
        if f32 == 2 and self._solver() == "cholesky" and self.diagonalShift < F32_BLOCK_MIN_SHIFT:
            # fp32 block sums perturb the scaled Gram by ~2e-8 per entry, i.e. ~2e-8 * 2 sqrt(P) in norm: the sum is no longer
            # exactly positive semi-definite, and a smaller shift cannot make the (rank-deficient) S + shift*diag(S) definite
            raise ValueError(f"f32BlockGram needs diagonalShift >= {F32_BLOCK_MIN_SHIFT:g} with the Cholesky solver "
                             f"(got {self.diagonalShift:g}); use the widening fp32 path (f32BlockGram=False) for smaller shifts")
        with self._Section(self, "gram"):             # bench.py: HIP events on the launch stream around the S-build
            if self.int8Gram and f32 in (0, 1) and self._int8_still_ok():
                # six digits for an fp64 Jacobian (47 bits); four for an fp32 one (31 bits: every stored value to 2^-31 of its row maximum)
                ns = self._i8_digits = 6 if f32 == 0 else 4
                need8 = nat.lib().vmcpde_gram_i8_workspace_bytes(P, n, ns, 8)
                if b.get("gram_i8_ws") is None or b["gram_i8_ws"].numel() * 8 < need8:
                    b["gram_i8_ws"] = torch.empty(need8 // 8 + 1, dtype=global_defs.tReal, device=JT.device)
                    b["i8_stats"] = torch.zeros(1, dtype=global_defs.tReal, device=JT.device)
                nat.call("vmcpde_gram_i8", nat.ptr(JT), ldj, f32, P, n, ns, 8, nat.ptr(c), nat.ptr(w), nat.ptr(out), nat.ptr(b["i8_stats"]),
                         nat.ptr(b["gram_i8_ws"]), b["gram_i8_ws"].numel() * 8, nat.stream())
                # heavy-tail figure of this call (largest row max / row rms) for the host, without synchronising: read before the next call
                if self._i8_host is None:
                    self._i8_host = torch.empty(1, dtype=global_defs.tReal).pin_memory()
                    self._i8_event = torch.cuda.Event()
                self._i8_host.copy_(b["i8_stats"], non_blocking=True)
                self._i8_event.record()
            else:
                nat.call("vmcpde_gram", nat.ptr(JT), ldj, f32, P, n, nat.ptr(c), nat.ptr(w), nat.ptr(out),
                         nat.ptr(b["gram_ws"]), b["gram_ws"].numel() * 8, nat.stream())

    # ---- tdvp.py:36-52 -------------------------------------------------------------------------------------------
    def get_tdvp_equation(self, Eloc, gradients, logProbs, psi=None):
        """Means, F0, S0 (and lazily S, SExp) from this rank's samples, reduced over ranks.  Returns F; the reference's
        other return values are the attribute `S` and the N x P array E~O~, which is never materialised here."""
        device = Eloc.device
        e = Eloc.reshape(-1).contiguous()
        n = e.numel()
        P = gradients.shape[-1]
        b = self._buffers(P, device)
        jt = psi.jt_buffer_of(gradients) if psi is not None and hasattr(psi, "jt_buffer_of") else None
        if jt is None:   # generic input: repack once into the padded parameter-major layout the kernels consume
            ldj = _round_up(n, 16)
            JT = torch.zeros((b["Ppad"], ldj), dtype=gradients.dtype if gradients.dtype == torch.float32 else global_defs.tReal, device=device)
            JT[:P, :n] = gradients.reshape(n, P).T
        else:
            JT, ldj = jt
        Ntot = float(mpi.globNumSamples) if mpi.globNumSamples else float(n * mpi.get_comm_size())
        world = mpi.get_comm_size()
        f32 = 1 if JT.dtype == torch.float32 else 0
        use_chol = self._solver() == "cholesky"
        # per-call results (fresh small tensors, so that attributes read after a later call keep their values)
        scal = torch.empty(8, dtype=global_defs.tReal, device=device)
        mean = torch.empty(P, dtype=global_defs.tReal, device=device)
        with torch.cuda.device(device):
            # health record of the previous solve (pinned pivots, residual) and the previous call's mean^2/var check: their
            # copies precede this call's sampling and flow kernels in the stream, so waiting for them here does not idle the GPU,
            # and it makes the choices below identical on every rank (the packed sums must agree)
            self._check_guard("previous step")
            shift_mode = self.gramShift
            if shift_mode == "auto":
                if self._ratio_event is not None:
                    self._ratio_event.synchronize()
                    self._uncentred_ok = bool(self._ratio_host[0] < 1e3)
                shift_mode = "uncentred" if self._uncentred_ok else "centred"
            # Uncentred fp64 Gram with two spare padded rows: E - cE and 1 are appended to the Jacobian as rows P and P+1, so the
            # syrk's last tile row carries sum_n O (E - cE) and sum_n O and no separate pass over J is needed (uncentred mode
            # already requires mean^2/var < 1e3, which also bounds the cancellation in sum O - N c)
            bordered = (self.fuseMoments and shift_mode == "uncentred" and JT.dtype == global_defs.tReal and JT.shape[0] >= P + 2
                        and P + 2 <= b["Ppad"])
            rows = (nat.ptr(JT[P]), nat.ptr(JT[P + 1])) if bordered else (None, None)
            mode = self.centering
            if mode == "auto":
                mode = "exact" if (self._c_prev is None or self._c_prev.numel() != P) else "lagged"

            def prepare(cE, with_rows):
                nat.call("vmcpde_tdvp_prepare", nat.ptr(e), n, nat.ptr(cE), *(rows if with_rows else (None, None)), nat.ptr(b["sums"]),
                         nat.ptr(b["count"]), nat.ptr(b["stats"][3:]), nat.ptr(b["prep_ws"]), b["prep_ws"].numel() * 8, nat.stream())

            if mode == "exact":
                # pass 1: raw sums -> global means (tdvp.py:37,41); tiny all-reduce when sharded
                prepare(None, False)
                nat.call("vmcpde_row_moments", nat.ptr(JT), ldj, f32, P, n, nat.ptr(e), None, None, nat.ptr(b["rowsum"]),
                         nat.ptr(b["rowdot"]), nat.stream())
                if world > 1:
                    small = torch.cat([b["rowsum"], b["sums"][:1], b["count"]])
                    mpi.allreduce_sum_(small)
                    c, cE = small[:P] / small[P + 1], small[P:P + 1] / small[P + 1]
                else:
                    c, cE = b["rowsum"] / n, b["sums"][:1] / n
                if bordered:
                    prepare(cE, True)
            else:       # the previous call's global means are the shift of the one-pass moments (no copies: the buffers are per call)
                c, cE = self._c_prev, self._cE_prev
                prepare(cE, bordered)
            c_gram = c if shift_mode == "centred" else None
            # pass 2 (tdvp.py:40-45 in shifted form): sum (O - c), sum (O - c)(E - cE); and the syrk (tdvp.py:46)
            if not bordered:
                nat.call("vmcpde_row_moments", nat.ptr(JT), ldj, f32, P, n, nat.ptr(e), nat.ptr(c), nat.ptr(cE), nat.ptr(b["rowsum"]),
                         nat.ptr(b["rowdot"]), nat.stream())
            self._gram(b, JT, ldj, P + 2 if bordered else P, n, c_gram, None, b["packed"])
            with self._Section(self, "allreduce"):
                mpi.allreduce_sum_(b["pack"])        # the one collective of the step (no-op on a single rank)
            self._replicated = self._Section(self, "replicated")      # finalize + factorisation + substitution + diagnostics: closed in solve()
            self._replicated.__enter__()
            if use_chol and b["A"] is None:
                b["A"] = torch.zeros((b["Ppad"], b["Ppad"]), dtype=global_defs.tReal, device=device)
            # the bordered Cholesky (row P of A = F, corner = a huge pivot) gets its border row from the same kernel
            chol_border = use_chol and P < b["Ppad"]
            mask = self._frozen_mask(P, device) if self.frozenParameters is not None else None
            nat.call("vmcpde_tdvp_moments", nat.ptr(b["packed"]), P, 1 if bordered else 0, nat.ptr(b["rowsum"]), nat.ptr(b["rowdot"]),
                     nat.ptr(b["sums"]), Ntot, nat.ptr(c), nat.ptr(c_gram), nat.ptr(cE), nat.ptr(mask), nat.ptr(b["F"]), nat.ptr(mean),
                     nat.ptr(b["diagS"]), nat.ptr(scal), float(self.svdTol) if use_chol else 0.0,
                     nat.ptr(b["A"][P]) if chol_border else None, _BORDER_PIVOT, nat.stream())
            # absolute floor svdTol * max diag(S) (scal[6]): the Tikhonov counterpart of the reference's spectral cut-off at
            # svdTol * lambda_max (tdvp.py:82-85).  The relative shift alone leaves parameters whose gradient is ~0 (S_ii ~ 1e-18 at
            # the reference's 1e-5 initialisation) with updates ~ 1/sqrt(S_ii), which no time step survives.
            nat.call("vmcpde_gram_finalize", nat.ptr(b["packed"]), nat.ptr(b["rowsum"]), None, None, nat.ptr(c), nat.ptr(c_gram), None, P, Ntot,
                     float(self.diagonalShift), nat.ptr(scal[6:]) if use_chol else None, nat.ptr(b["S"]), b["Ppad"],
                     nat.ptr(b["A"]) if use_chol else None, b["Ppad"], None, None, nat.stream())
            if self.frozenParameters is not None:
                idx = self._frozen_index(device)
                b["S"][idx, :] = 0.0
                b["S"][:, idx] = 0.0
                b["F"][idx] = 0.0
                b["diagS"][idx] = 0.0
                if use_chol:
                    b["A"][idx, :P] = 0.0
                    b["A"][:P, idx] = 0.0
                    b["A"][idx, idx] = scal[6]         # the floor alone: a positive pivot, update exactly 0 (F is 0 there)
                    if chol_border:
                        b["A"][P, idx] = 0.0
            if self.gramShift == "auto":     # asynchronous safety check for the next call: no host sync here
                if self._ratio_host is None:
                    self._ratio_host = torch.empty(1, dtype=global_defs.tReal).pin_memory()
                    self._ratio_event = torch.cuda.Event()
                self._ratio_host.copy_(scal[4:5], non_blocking=True)
                self._ratio_event.record()
        self.ElocMean, self.ElocMeanAbs, self.ElocVar = scal[0], scal[1], scal[2]     # tdvp.py:37-39
        self._scal = scal
        self._floor = scal[6] if (use_chol and self.svdTol > 0) else None
        self._c_prev, self._cE_prev = mean, scal[0:1]
        self.F0 = b["F"][:P]
        self.S0 = b["S"][:P, :P]
        # SExp / rhoVar / snr are evaluated lazily from the retained Jacobian and log-probabilities, which live in psi's per-N
        # workspace (2 GB at C2: not copied): its generation is recorded here and checked on use.
        self._lazy = dict(JT=JT, ldj=ldj, P=P, n=n, Ntot=Ntot, e=e, logp=logProbs.reshape(-1), b=b, mean=mean, Emean=scal[0],
                          chol_border=chol_border,
                          owner=psi if jt is not None else None,
                          gen=psi.workspace_generation(n) if jt is not None else None)
        self._SExp = self._rhoVar = self._snr = self._S = None
        return self.F0

    def _frozen_mask(self, P, device):
        m = getattr(self, "_frozen_mask_t", None)
        if m is None or m.device != device or m.numel() != P:
            m = torch.ones(P, dtype=global_defs.tReal, device=device)
            m[self._frozen_index(device)] = 0.0
            self._frozen_mask_t = m
        return m

    def _frozen_index(self, device):
        idx = getattr(self, "_frozen_idx", None)
        if idx is None or idx.device != device:
            idx = self._frozen_idx = torch.as_tensor(np.asarray(self.frozenParameters, dtype=np.int64), device=device)
        return idx

    def _weighted_gram(self, w):
        """(1/N) sum_n w_n^2 O~_n O~_n^T over all ranks from the retained Jacobian (w = logp: tdvp.py:33,47; w = E~: the
        second moment behind rhoVar, tdvp.py:68-70; w = None: S0 itself, for the `S0` attribute after a matrix-free call)."""
        z = self._lazy
        if z["owner"] is not None and z["owner"].workspace_generation(z["n"]) != z["gen"]:
            raise RuntimeError("SExp / rhoVar / snr are computed lazily from the Jacobian of the last TDVP call, but the VarState "
                               "has evaluated another batch of the same size since (its workspace was overwritten): read these "
                               "attributes before calling psi(...), psi.hessian or psi.evaluate again")
        P, device = z["P"], z["JT"].device
        b = z["b"] if z["b"] is not None else self._buffers(P, device)     # (a matrix-free call has made none)
        tmp = torch.zeros(b["npk"], dtype=global_defs.tReal, device=device)
        out = torch.zeros((b["Ppad"], b["Ppad"]), dtype=global_defs.tReal, device=device)
        zero = torch.zeros(P, dtype=global_defs.tReal, device=device)
        with torch.cuda.device(device):
            self._gram(b, z["JT"], z["ldj"], P, z["n"], z["mean"], None if w is None else w.contiguous(), tmp)
            mpi.allreduce_sum_(tmp)
            nat.call("vmcpde_gram_finalize", nat.ptr(tmp), nat.ptr(zero), None, None, None, None, None, P, z["Ntot"], 0.0, None, nat.ptr(out),
                     b["Ppad"], None, 0, None, None, nat.stream())
        if self.frozenParameters is not None:
            idx = self._frozen_index(device)
            out[idx, :] = 0.0
            out[:, idx] = 0.0
        return out[:P, :P]

    @property
    def S0(self):
        """tdvp.py:46.  After a matrix-free call (which never assembles it) it is formed on first access from the retained Jacobian."""
        if self._S0 is None and getattr(self, "_S0_lazy", False):
            self._S0 = self._weighted_gram(None)
            self._S0_lazy = False
        return self._S0

    @S0.setter
    def S0(self, value):
        self._S0 = value
        self._S0_lazy = False

    @property
    def S(self):
        """tdvp.py:48-51: S0, or S0 + diagonalShift*diag(S0)."""
        if self._S is None:
            self._S = self.S0 if not (self.diagonalShift > 1e-10) else self.S0 + torch.diag(self.diagonalShift * torch.diag(self.S0))
        return self._S

    @property
    def SExp(self):
        """tdvp.py:47: <(logp O~)(logp O~)^T>, consumed by AdaptiveHeun (stepper.py:71)."""
        if self._SExp is None:
            self._SExp = self._weighted_gram(self._lazy["logp"])
        return self._SExp

    def SExp_operator(self):
        """What the steppers hand to the norm function in place of the matrix `SExp`: a GramOperator where the quadratic form can be
        taken from the retained fp64 Jacobian in one pass, the (lazily formed) matrix itself otherwise or once it exists."""
        z = getattr(self, "_lazy", None)
        if self._SExp is not None or z is None or z.get("owner") is None or z["JT"].dtype != global_defs.tReal or \
                os.environ.get("VMCPDE_SEXP_OPERATOR", "1") == "0":
            return self.SExp
        return GramOperator(self)

    def _sexp_pass(self, v):
        """q_n = logp_n (J_n - m).v for this rank's samples (first pass of `vmcpde_jt_matvec`), and the lazily kept record."""
        z = self._lazy
        if z["owner"] is not None and z["owner"].workspace_generation(z["n"]) != z["gen"]:
            raise RuntimeError("SExp / rhoVar / snr are computed lazily from the Jacobian of the last TDVP call, but the VarState "
                               "has evaluated another batch of the same size since (its workspace was overwritten): read these "
                               "attributes before calling psi(...), psi.hessian or psi.evaluate again")
        P, n, device = z["P"], z["n"], z["JT"].device
        v = v.to(device=device, dtype=global_defs.tReal).contiguous()
        if self.frozenParameters is not None:
            v = v * self._frozen_mask(P, device)
        need = nat.lib().vmcpde_jt_matvec_workspace_bytes(P, n)
        w = getattr(self, "_sexp_ws", None)
        if w is None or w[0].numel() * 8 < need or w[1].numel() < n or w[0].device != device:
            w = self._sexp_ws = (torch.empty(need // 8, dtype=global_defs.tReal, device=device), torch.empty(n, dtype=global_defs.tReal, device=device))
        with torch.cuda.device(device):
            nat.call("vmcpde_jt_matvec", nat.ptr(z["JT"]), z["ldj"], P, n, nat.ptr(v), nat.ptr(w[1]), None, nat.ptr(w[0]), w[0].numel() * 8, nat.stream())
        return z["logp"].reshape(-1) * (w[1][:n] - torch.dot(z["mean"], v)), v

    def _sexp_form(self, v):
        """v^T SExp v = (1/N) sum_n q_n^2 (tdvp.py:47 with the centred gradients of :41-42), summed over the ranks."""
        q, _ = self._sexp_pass(v)
        tot = torch.dot(q, q).reshape(1)
        mpi.allreduce_sum_(tot)
        return tot[0] / self._lazy["Ntot"]

    def _sexp_matvec(self, v):
        """SExp v = (1/N) sum_n logp_n q_n (J_n - m): a second pass with the weights folded into z."""
        z = self._lazy
        q, vv = self._sexp_pass(v)
        g = (z["logp"].reshape(-1) * q).contiguous()
        Jv = z["JT"][:z["P"], :z["n"]]
        y = Jv @ g - z["mean"] * g.sum()
        mpi.allreduce_sum_(y)
        y = y / z["Ntot"]
        if self.frozenParameters is not None:
            y = y * self._frozen_mask(z["P"], y.device)
        return y

    @property
    def rhoVar(self):
        """tdvp.py:68-70: Var_n[(E~ O~ V)_nk] = diag(V^T G V) - VtF^2 with G = <E~^2 O~ O~^T> (no N x P GEMM)."""
        if self._rhoVar is None:
            if self.V is None:
                raise RuntimeError("rhoVar/snr need the eigenbasis: use solver='eigh'")
            G = self._weighted_gram(self._lazy["e"] - self._lazy["Emean"])
            self._rhoVar = ((G @ self.V) * self.V).sum(0) - self.VtF ** 2
        return self._rhoVar

    @property
    def snr(self):
        """tdvp.py:71."""
        if self._snr is None:
            if self.V is None:
                raise RuntimeError("rhoVar/snr need the eigenbasis: use solver='eigh'")
            self._snr = torch.sqrt(torch.abs(self._lazy["Ntot"] * (self.VtF * self.VtF) / self.rhoVar))
        return self._snr

    def _solver(self):
        return "eigh" if self.solver == "auto" else self.solver

    # ---- Cholesky health: pinned pivots and the solver residual ------------------------------------------------------
    def _record_guard(self, out, chol):
        """Queue (residual, pinned-pivot count) of this solve for the host without synchronising: checked by `_check_guard` at
        the next host-side guard (the NaN check of `__call__`, `check()` -- called by the steppers before a step is accepted --, or
        the start of the next right-hand side)."""
        self._guard_chol = chol
        if self._guard_host is None:
            self._guard_host = torch.empty(4, dtype=global_defs.tReal).pin_memory()
            self._guard_event = torch.cuda.Event()
        self._guard_host.copy_(out, non_blocking=True)
        self._guard_event.record()
        self._guard_pending = True

    def check(self, when="this step"):
        """Wait for the health record of the last solve and raise if it failed (the reference's NaN exit, tdvp.py:136-141, and the
        Cholesky guard).  `__call__` runs it itself when observables are on; with `observables=False` nothing synchronises with the
        host inside a call, so the steppers call it before they hand a step back and a driver should call it before writing a
        checkpoint -- otherwise a failed solve would only be noticed at the start of the next right-hand side (or never, for the
        last call of a run)."""
        self._check_guard(when)

    def _check_guard(self, when):
        """factor_inverse_32 (csrc/solve.hip) replaces every non-positive / NaN pivot by a unit row and only counts it, so an
        indefinite or NaN-contaminated S gives a finite but wrong update: fail loudly instead of integrating on.  On the spectral
        path the record holds the residual |S u - F| / |F|, which is NaN exactly when the update (or S, F) is."""
        if not getattr(self, "_guard_pending", False):
            return
        self._guard_event.synchronize()
        self._guard_pending = False
        resid, pinned = float(self._guard_host[0]), int(self._guard_host[2])
        if not self._guard_chol:
            if math.isnan(resid):
                raise FloatingPointError(f"nan encountered in the TDVP update ({when})")
            return
        if not (resid <= self.residualBound):      # also catches NaN
            if (self.choleskyInfo or {}).get("mode") == "static":
                # the rank of S outgrew the plan of the low-rank solve within one call (it is re-planned 32 columns before the pad, and a
                # factorisation that stops at the pad still refines to ~1e-8: this needs a jump of the rank by a large factor): the
                # plan is dropped, so a caller that catches this and repeats the step gets a planning call
                self._chol_static = None
                self._mf_plan = None
                raise StaticPlanOutdated(f"low-rank Cholesky solve ({when}): solver residual {resid:.3e} > {self.residualBound:g} -- the "
                                         f"numerical rank of S outgrew the static plan {self.choleskyInfo}; the next call plans anew "
                                         f"(choleskyEngine='dense' avoids the low-rank route)")
            raise FloatingPointError(f"on-device Cholesky failed ({when}): solver residual {resid:.3e} > {self.residualBound:g} with "
                                     f"{pinned} pinned pivot(s); S + shift is not positive definite -- raise diagonalShift / svdTol "
                                     f"or use solver='eigh'")
        if pinned and not getattr(self, "_warned_pinned", False):
            import warnings
            warnings.warn(f"Cholesky pinned {pinned} parameter(s) with a non-positive pivot (dead or padded parameters; their update "
                          f"is 0); residual {resid:.2e} is within bound", RuntimeWarning)
            self._warned_pinned = True

    # ---- the pivoted Cholesky factor of S shared by the rank-revealing spectral decomposition and the low-rank Cholesky solve ----
    def _lowrank_buffers(self, P, device, cap):
        key = ("lowrank", P, str(device))
        w = self._buf_lr.get(key) if hasattr(self, "_buf_lr") else None
        if w is None or w["LT"].shape[0] < cap:
            ldl = _round_up(P, 64)
            w = dict(ldl=ldl,
                     LT=torch.empty((cap, ldl), dtype=global_defs.tReal, device=device),
                     YT=None, tau=None, hq=None,
                     piv=torch.zeros(cap, dtype=torch.int32, device=device),
                     state=torch.zeros(2, dtype=torch.int32, device=device),
                     ws=torch.zeros(nat.lib().vmcpde_pivoted_cholesky_blocked_workspace_bytes(P) // 8, dtype=global_defs.tReal, device=device),
                     state_host=torch.zeros(2, dtype=torch.int32).pin_memory(),
                     info_host=torch.zeros(3, dtype=global_defs.tReal).pin_memory(),
                     head_host=torch.zeros(4, dtype=global_defs.tReal).pin_memory())
            self._buf_lr = {key: w}
        return w

    def _lowrank_tol(self):
        """Pivoting tolerance of the low-rank routes for THIS call's S, or None when they do not apply: the factorisation must stop
        above the noise of S.  An fp64-accumulated Gram (fp64 or fp32 Jacobian storage) is exact to round-off: LOWRANK_TOL.  The
        six-digit int8 S-build is 1.2e-13 |S|_F away from it: 1e-12.  Four int8 digits (fp32 Jacobian: 6e-9) and fp32 block sums
        (2e-8) bury everything below ~1e-8 max diag in noise -- such an S has no usable numerical rank (measured: a plan made on the
        first, fp64-built S failed the residual guard on the first int8-built one), so those variants keep the dense paths."""
        if self.f32BlockGram:
            return None
        tol = LOWRANK_TOL
        if self.int8Gram and not self._i8_off:
            if self._i8_digits == 4:
                return None
            tol = 1e-12
        if self._uncentred_ok and self._ratio_host is not None:
            # the uncentred Gram subtracts N m m^T afterwards: relative noise ~ eps * max_p mean_p^2 / var_p (the figure the engine
            # watches to stay below 1e3; 1.3e-4 for the headline network), read from the previous call's record
            r = float(self._ratio_host[0])
            if r == r:
                tol = max(tol, 2e-15 * r)
        return tol

    def _arm_hints(self, w, P):
        """Before a factorisation starts in `w`: its previous pivots (still in w["piv"] / w["state"], zero on first use) become the
        hint list, or the list is disarmed."""
        on = bool(self.pivotHints) and os.environ.get("VMCPDE_PIVOT_HINTS", "1") != "0"
        nat.call("vmcpde_lowrank_hint", nat.ptr(w["piv"]) if on else None, nat.ptr(w["state"]) if on else None, P, nat.ptr(w["ws"]),
                 w["ws"].numel() * 8, nat.stream())

    @staticmethod
    def _steps_used(ws_head_host):
        """Block steps that added columns (PivWork::steps, csrc/eig.hip) from a host copy of the first four doubles of the workspace."""
        return int(ws_head_host.view(torch.int32)[7])

    def _pivoted_factor(self, S, w, cap, extra=None):
        """S = L L^T + R by blocked diagonal pivoting (up to 32 columns per pair of launches) into w["LT"]; returns (rank, converged,
        max diag, trace of R, extra's value).  The first call covers the previous call's rank (the spectrum moves slowly along a
        trajectory); the host reads `state` after every call (`extra`: a device scalar copied along)."""
        P, device = S.shape[0], S.device
        done, first = False, True
        self._arm_hints(w, P)
        # (first launch chain: what the previous factorisation in these buffers used plus a third; an unhinted step accepts ~13 columns)
        used = self._steps_used(w["head_host"])
        nblocks = used + max(3, used // 3) if used else ((self._lowrank_last_rank + 12) // 13 + 2 if self._lowrank_last_rank else 12)
        for _ in range(cap // 3 + 8):           # (every step makes progress -- a column, or candidates lowered to what is left of them)
            nat.call("vmcpde_pivoted_cholesky_blocked", nat.ptr(S), S.stride(0), P, self._lowrank_tol() or LOWRANK_TOL, 1 if first else 0, nblocks, cap, nat.ptr(w["LT"]),
                     w["ldl"], nat.ptr(w["piv"]), nat.ptr(w["state"]), nat.ptr(w["ws"]), w["ws"].numel() * 8, nat.stream())
            w["state_host"].copy_(w["state"], non_blocking=True)
            w["info_host"][:2].copy_(w["ws"][:2], non_blocking=True)
            w["head_host"].copy_(w["ws"][:4], non_blocking=True)
            if extra is not None and first:
                w["info_host"][2:3].copy_(extra.reshape(1), non_blocking=True)
            torch.cuda.current_stream(device).synchronize()
            first, nblocks = False, 3
            if int(w["state_host"][1]):
                done = True
                break
            if int(w["state_host"][0]) >= cap:
                break
        r = int(w["state_host"][0])
        done = done or r == P                                       # full rank: every row was a pivot, nothing is left
        return r, done, float(w["info_host"][0]), float(w["info_host"][1]), float(w["info_host"][2])

    # ---- solver="cholesky" through the low-rank factor ----------------------------------------------------------------
    def _solve_lowrank(self, b, F, P, update):
        """(S0 + D) u = F with D = diagonalShift diag(S0) + floor I > 0 and S0 = L L^T + R:  M = D + L L^T is inverted by the Woodbury
        identity, M^-1 v = D^-1 v - D^-1 L G^-1 L^T D^-1 v with G = I + L^T D^-1 L (r x r, Cholesky), and u <- u + M^-1 (F - (S0 + D) u)
        with the TRUE S0 removes what R (|M^-1 R| <= trace(R) / min D) and the identity's cancellation left.  False: not applicable
        (rank above the cap, D not positive, R too large) -- the caller factors the full matrix.

        Two modes.  PLANNING (first call, and whenever the plan no longer fits): the host reads the rank after the factorisation and
        sizes everything by it.  STATIC (the calls after): the rank is padded to the plan's rmax with zero columns of L -- G gets an
        identity block there --, the factorisation runs a fixed number of block steps (those after the criterion return at once), and
        the host reads nothing: `state` is copied asynchronously and looked at by the NEXT call, which re-plans if the factorisation
        did not finish or the rank came close to rmax.  A call whose factorisation stopped short still refines against the full
        matrix, and its solver residual is measured and guarded like every other (`residualBound`)."""
        device = F.device
        S0 = b["S"][:P, :P]
        cap = P if self.choleskyEngine == "lowrank" else max(1, min(LOWRANK_RANK_CAP, P // LOWRANK_RANK_DIVISOR))
        w = self._lowrank_buffers(P, device, cap)
        D = b["diagS"] * float(self.diagonalShift)
        if self._floor is not None:
            D = D + self._floor
        st = self._chol_static
        if st is not None and st["P"] == P and st["rmax"] <= w["LT"].shape[0]:
            if st["event"] is not None:                 # the previous static call's factorisation: finished, with room to spare?
                st["event"].synchronize()
                prev_rank, prev_done = int(st["host"][0]), int(st["host"][1])
                if not prev_done or prev_rank > st["rmax"] - 32:     # (re-plan well before the rank reaches the pad)
                    st = self._chol_static = None
                else:                                                # what the last factorisation used, and a third (at least 3) to spare
                    used = self._steps_used(st["head"])
                    st["nsteps"] = used + max(3, used // 3)
            if st is not None:
                self._solve_lowrank_static(b, F, P, update, w, D, st)
                return True
        r, done, d0, left, dmin = self._pivoted_factor(S0, w, cap, extra=D.min())
        rho = left / dmin if dmin > 0 else float("inf")
        self.choleskyInfo = dict(engine="lowrank", mode="planning", rank=r, cap=cap, max_diag=d0, trace_left=left, min_shift=dmin, converged=done)
        if not done or r < 1 or not (rho <= LOWRANK_SOLVE_RHO):
            return False
        self._lowrank_last_rank = r
        LTr = w["LT"][:r, :P]
        Ls = LTr / D                                                # L^T D^-1  (r x P)
        G = Ls @ LTr.T
        G.diagonal().add_(1.0)
        Gc = torch.linalg.cholesky(G)

        def minv(v):
            t = torch.cholesky_solve((Ls @ v)[:, None], Gc)[:, 0]
            return v / D - Ls.T @ t

        u = minv(F)
        Su = torch.empty(P, dtype=global_defs.tReal, device=device)
        nref = 1 if rho <= 1e-6 else 2             # (error after j refinements ~ rho^(j+1); the residual is measured either way)
        for _ in range(nref):
            nat.call("vmcpde_gemv", nat.ptr(b["S"]), b["Ppad"], P, P, 0, nat.ptr(u), nat.ptr(Su), nat.stream())
            u = u + minv(F - Su - D * u)
        update.copy_(u)
        b["info"].zero_()
        rmax = _round_up(r + 40, 128)
        if rmax <= w["LT"].shape[0] and rmax <= cap + 127:
            used = self._steps_used(w["head_host"])
            self._chol_static = dict(P=P, rmax=rmax, nsteps=used + max(3, used // 3), nref=nref, rank=r, event=None,
                                     host=torch.zeros(2, dtype=torch.int32).pin_memory(),
                                     head=torch.zeros(4, dtype=global_defs.tReal).pin_memory())
        return True

    def _solve_lowrank_static(self, b, F, P, update, w, D, st):
        device, rmax = F.device, st["rmax"]
        S0 = b["S"][:P, :P]
        if w.get("G") is None or w["G"].shape[0] != rmax:
            need = nat.lib().vmcpde_chol_workspace_bytes(rmax)
            w["G"] = torch.zeros((rmax, rmax), dtype=global_defs.tReal, device=device)
            w["G_ws"] = torch.empty(need // 8, dtype=global_defs.tReal, device=device)
            w["G_info"] = torch.zeros(1, dtype=torch.int32, device=device)
            w["t2"] = torch.empty(rmax, dtype=global_defs.tReal, device=device)
            w["Su"] = torch.empty(P, dtype=global_defs.tReal, device=device)
        G, gws, t2, Su = w["G"], w["G_ws"], w["t2"], w["Su"]
        need = gws.numel() * 8
        w["LT"][:rmax].zero_()                                      # columns past the rank: zero, i.e. an identity block in G
        self._arm_hints(w, P)                                       # (rmax is the padded size of G; a small P caps the rank below it)
        nat.call("vmcpde_pivoted_cholesky_blocked", nat.ptr(S0), S0.stride(0), P, self._lowrank_tol() or LOWRANK_TOL, 1, st["nsteps"], min(rmax, P), nat.ptr(w["LT"]), w["ldl"],
                 nat.ptr(w["piv"]), nat.ptr(w["state"]), nat.ptr(w["ws"]), w["ws"].numel() * 8, nat.stream())
        st["host"].copy_(w["state"], non_blocking=True)             # for the next call
        st["head"].copy_(w["ws"][:4], non_blocking=True)
        if st["event"] is None:
            st["event"] = torch.cuda.Event()
        st["event"].record()
        LTr = w["LT"][:rmax, :P]
        # G = I + L^T D^-1 L, contraction length P, result only rmax x rmax: as ONE library GEMM the 256 x 256 x 4213 shape runs on 8
        # workgroups (0.23 ms; the engine's own weighted syrk path 0.17 ms) -- as a batched GEMM over column chunks of 128 (the zero pad
        # of the rows of LT included) it fills the chip, and the chunk sums are added up
        ldl = w["ldl"]
        ck = 128 if ldl % 128 == 0 else 64
        if w.get("Dpad") is None:
            w["Dpad"] = torch.ones(ldl, dtype=global_defs.tReal, device=device)
        w["Dpad"][:P] = D
        Lfull = w["LT"][:rmax]                                      # rmax x ldl, pad columns zero
        A3 = (Lfull / w["Dpad"]).view(rmax, ldl // ck, ck).permute(1, 0, 2)
        B3 = Lfull.view(rmax, ldl // ck, ck).permute(1, 2, 0)
        torch.sum(torch.bmm(A3, B3), dim=0, out=G)
        G.diagonal().add_(1.0)
        nat.call("vmcpde_cholesky", nat.ptr(G), rmax, rmax, nat.ptr(w["G_info"]), nat.ptr(gws), need, nat.stream())

        def minv(v):
            y = v / D
            t1 = LTr @ y
            nat.call("vmcpde_chol_solve", nat.ptr(G), rmax, rmax, nat.ptr(t1), nat.ptr(t2), nat.ptr(gws), need, nat.stream())
            return torch.addcdiv(y, LTr.T @ t2, D, value=-1.0)

        u = minv(F)
        for _ in range(st["nref"]):
            nat.call("vmcpde_gemv", nat.ptr(b["S"]), b["Ppad"], P, P, 0, nat.ptr(u), nat.ptr(Su), nat.stream())
            u = u + minv(F - Su - D * u)
        update.copy_(u)
        b["info"].zero_()
        self.choleskyInfo = dict(engine="lowrank", mode="static", rmax=rmax, block_steps=st["nsteps"], refinements=st["nref"],
                                 planned_rank=st["rank"])

    # ---- tdvp.py:57-64 for a numerically rank-deficient S -----------------------------------------------------------
    def _eigh_lowrank(self, S):
        """(ev ascending, V) of the symmetric positive semi-definite S from a pivoted Cholesky factor (csrc/eig.hip), or None when the
        numerical rank exceeds the cap (the caller then takes the dense path).  S = L L^T + R with trace(R) <= (P - r) LOWRANK_TOL max diag;
        L = Q1 R1 by Householder reflectors, Q = [Q1 | Q2] = I - Y T Y^T; Q1^T S Q1 = W diag(w) W^T; ev = sort([w, 0 ... 0]),
        V = [Q1 W | Q2] in the same order.  What is not in csrc/eig.hip is library GEMM / TRSM and the r x r decomposition."""
        P, device = S.shape[0], S.device
        # (forced engine: no cap -- the factorisation runs until its criterion is met)
        cap = P if self.spectralEngine == "lowrank" else max(1, min(LOWRANK_RANK_CAP, P // LOWRANK_RANK_DIVISOR))
        w = self._lowrank_buffers(P, device, cap)
        # (a NaN on the diagonal compares as "smaller" in the pivot search and would simply be left out: the trace rides along with the
        # first host read, and a non-finite S goes to the dense path, which reports it as before)
        r, done, d0, left, tr = self._pivoted_factor(S, w, cap, extra=torch.diagonal(S).sum())
        self.spectralInfo = dict(engine="lowrank", rank=r, cap=cap, max_diag=d0, trace_left=left, converged=done)
        if not done or r < 1 or not (left <= LOWRANK_TRACE_BOUND * d0) or not np.isfinite(tr):
            return None
        self._lowrank_last_rank = r
        Y, Xn, Q1 = self._lowrank_range_basis(w, P, r, device)
        SQ = S @ Q1
        T = _long_k_product(Q1.T.contiguous(), SQ.T.contiguous())
        T = 0.5 * (T + T.T)
        # (the r x r core on one host LAPACK thread takes 1.6 ms against 5.4 ms for syevd's launch chain at r = 238, but the host
        # synchronisation it needs costs the calling loop more than that: 33.0 against 30.2 ms per TDVP call -- measured and dropped)
        wv, W = torch.linalg.eigh(T)                               # r x r (rocSOLVER)
        U = Q1 @ W
        return self._lowrank_assemble(Y, Xn, U, wv, r, P, device)

    @staticmethod
    def _lowrank_assemble(Y, Xn, U, wv, r, P, device):
        """(ev, V) from the Ritz pairs (wv ascending, U = Q1 W) and the compact WY form of Q = [Q1 | Q2] = I + Y^T Xn."""
        # ascending order: the (few, round-off) negative Ritz values, the P - r zeros of the complement, the positive Ritz values.  The
        # complement Q2 = E_2 - Y T Y_2^T is written by its GEMM straight into its column block of V: one pass over P x P instead of
        # four (GEMM, sign, diagonal, column gather: 7 + 2 of the 29 ms of the engine at P = 16 181)
        nneg = int((wv < 0).sum())
        V = torch.empty((P, P), dtype=global_defs.tReal, device=device)
        V[:, :nneg] = U[:, :nneg]
        V[:, nneg + P - r:] = U[:, nneg:]
        if P > r:
            Vn = V[:, nneg:nneg + P - r]
            torch.mm(Y.T, Xn[:, r:], out=Vn)
            Vn[r:].diagonal().add_(1.0)
        ev = torch.cat([wv[:nneg], torch.zeros(P - r, dtype=global_defs.tReal, device=device), wv[nneg:]])
        return ev, V

    def _lowrank_range_basis(self, w, P, r, device):
        """Householder QR of the factor in w["LT"][:r] (destroyed): reflectors Y (r x P), Xn = -T Y^T of the compact WY form
        Q = I + Y^T Xn, and Q1 = Q[:, :r], an orthonormal basis of range(L)."""
        if w["YT"] is None or w["YT"].shape[0] < r:
            w["YT"] = torch.empty((w["LT"].shape[0], w["ldl"]), dtype=global_defs.tReal, device=device)
            w["tau"] = torch.empty(w["LT"].shape[0], dtype=global_defs.tReal, device=device)
            w["hq"] = torch.empty(5 * w["LT"].shape[0] + 5, dtype=global_defs.tReal, device=device)
        nat.call("vmcpde_householder_rows", nat.ptr(w["LT"]), w["ldl"], P, r, nat.ptr(w["YT"]), nat.ptr(w["tau"]), nat.ptr(w["hq"]), nat.stream())
        Y = w["YT"][:r, :P]                                        # r x P: row k = reflector k
        Yp = w["YT"][:r]                                           # (r x ldl: the pad columns are zero)
        Tinv = torch.triu(_long_k_product(Yp, Yp), 1)
        Tinv.diagonal().copy_(1.0 / w["tau"][:r])                  # T^-1 = strict upper part of Y^T Y + diag(1 / tau)
        # T itself (r x r, what dlarft forms), then T Y^T as a GEMM: a triangular solve with P right-hand sides makes rocBLAS allocate
        # a workspace that fails for r in the thousands
        Tm = torch.linalg.solve_triangular(Tinv, torch.eye(r, dtype=global_defs.tReal, device=device), upper=True)
        Xn = -(Tm @ Y)                                             # -T Y^T  (r x P);  Q = I - Y T Y^T, columns [Q1 | Q2]
        Q1 = Y.T @ Xn[:, :r]
        Q1[:r].diagonal().add_(1.0)                                # Q1 = E_1 - Y T Y_1^T  (P x r, orthonormal basis of range(L))
        return Y, Xn, Q1

    # ---- tdvp.py:57-71 -------------------------------------------------------------------------------------------
    def transform_to_eigenbasis(self, S, F, EOdata=None):
        P = F.numel()
        done = False
        on_device = (P >= DEVICE_EIGH_MIN_P) if self.diagonalizeOnDevice is None else bool(self.diagonalizeOnDevice)
        if self.spectralEngine == "lowrank" and self.diagonalizeOnDevice is None:
            on_device = True
        self.spectralInfo = None
        lowrank = on_device and self.spectralEngine != "dense" and not (self.diagonalShift > 1e-10) and self._lowrank_tol() is not None and \
            (self.spectralEngine == "lowrank" or (P >= LOWRANK_MIN_P and self._lowrank_backoff == 0))
        if on_device and self.spectralEngine == "auto" and self._lowrank_backoff > 0:
            self._lowrank_backoff -= 1
        if lowrank:
            with torch.cuda.device(F.device):
                try:
                    res = self._eigh_lowrank(S)
                except RuntimeError as exc:       # e.g. rocBLAS cannot allocate the TRSM workspace for a rank in the thousands (forced engine)
                    import warnings
                    warnings.warn(f"rank-revealing decomposition failed ({str(exc)[:80]}); using the dense eigh for this call", RuntimeWarning)
                    res = None
                    self.spectralInfo = dict(self.spectralInfo or {}, engine="lowrank", converged=False, error=str(exc)[:80])
            if res is not None:
                self.ev, self.V = res
                done = True
            elif self.spectralEngine == "auto":
                self._lowrank_backoff = LOWRANK_RETRY_AFTER        # rank above the cap: dense for the next calls
        if on_device and not done:
            if self.spectralInfo is None or self.spectralInfo.get("engine") != "lowrank":
                self.spectralInfo = dict(engine="dense")
            else:
                self.spectralInfo = dict(self.spectralInfo, engine="dense", tried="lowrank")
            try:
                self.ev, self.V = torch.linalg.eigh(S)                        # tdvp.py:59 (rocSOLVER)
                done = True
            except RuntimeError as exc:                                       # rocSOLVER did not converge on this (singular) S
                import warnings
                warnings.warn(f"rocSOLVER eigh failed ({str(exc)[:80]}); using host LAPACK for this call, as the reference does "
                              f"with diagonalizeOnDevice=False (tdvp.py:61-64)", RuntimeWarning)
                done = False
        if not done:
            with _lapack_threads(P):
                tmpEv, tmpV = np.linalg.eigh(S.detach().cpu().numpy())        # tdvp.py:61-64 (host LAPACK)
            self.ev = torch.from_numpy(tmpEv).to(F.device)
            self.V = torch.from_numpy(tmpV).to(F.device)
        self.VtF = torch.empty(P, dtype=global_defs.tReal, device=F.device)
        Vc = self.V.contiguous()
        self.V = Vc
        with torch.cuda.device(F.device):
            nat.call("vmcpde_gemv", nat.ptr(Vc), P, P, P, 1, nat.ptr(F.contiguous()), nat.ptr(self.VtF), nat.stream())   # tdvp.py:66

    # ---- EXPERIMENT: the Cholesky-path step without an assembled S ----------------------------------------------------
    def _solve_matrix_free(self, Eloc, gradients, logProbs, psi):
        """(S0 + D) u = F as in `_solve_lowrank`, with S0 = J^T J / N - m m^T touched only through its diagonal, the columns the pivoted
        factorisation asks for and products S0 v.  First call: planning (the host reads the rank and the number of block steps used);
        afterwards a static plan, checked one call later (as `_solve_lowrank_static`)."""
        world = mpi.get_comm_size()      # sharded samples: the O(P) moments, every 32-column panel and every S0 v are summed over the ranks
        spectral = self._solver() != "cholesky"
        if spectral and (self.diagonalShift > 1e-10 or self.useSNR):
            raise NotImplementedError("matrixFree with the spectral solve: diagonalShift = 0 and useSNR = False only (a shifted S is not rank "
                                      "deficient; the SNR needs the E~-weighted Gram)")
        jt = psi.jt_buffer_of(gradients) if psi is not None and hasattr(psi, "jt_buffer_of") else None
        if jt is None or jt[0].dtype != global_defs.tReal:
            raise NotImplementedError("matrixFree needs the engine's own fp64 parameter-major Jacobian buffer")
        JT, ldj = jt
        e = Eloc.reshape(-1).contiguous()
        n, P, device = e.numel(), gradients.shape[-1], e.device
        Jv = JT[:P, :n]
        tol = self._lowrank_tol() or LOWRANK_TOL
        with torch.cuda.device(device):
            self._check_guard("previous step")
            cap = min(LOWRANK_RANK_CAP, max(P // LOWRANK_RANK_DIVISOR, min(P, 256)))      # (small problems: up to 256 columns)
            w = self._lowrank_buffers(P, device, _round_up(cap + 40, 128))                  # (rows for the plan's zero padding)
            ldl = w["ldl"]
            if w.get("Scol") is None:
                w["Scol"] = torch.zeros((32, ldl), dtype=global_defs.tReal, device=device)
                need = nat.lib().vmcpde_gram_columns_workspace_bytes(P, n)
                w["gc_ws"] = torch.empty(need // 8, dtype=global_defs.tReal, device=device)
                w["rowsum"] = torch.empty(P, dtype=global_defs.tReal, device=device)
                w["rowdot"] = torch.empty(P, dtype=global_defs.tReal, device=device)
                w["rowsq"] = torch.empty(P, dtype=global_defs.tReal, device=device)
            if w["gc_ws"].numel() * 8 < nat.lib().vmcpde_gram_columns_workspace_bytes(P, n):
                w["gc_ws"] = torch.empty(nat.lib().vmcpde_gram_columns_workspace_bytes(P, n) // 8, dtype=global_defs.tReal, device=device)
            with self._Section(self, "gram"):          # (what replaces the S-build: moments, diagonal, and the factorisation's passes over J)
                Ntot = float(mpi.globNumSamples) if mpi.globNumSamples else float(n * world)
                esums = torch.stack([e.sum(), (e * e).sum(), e.abs().sum()])
                if world > 1:
                    mpi.allreduce_sum_(esums)
                Em, E2, Eabs = esums[0] / Ntot, esums[1] / Ntot, esums[2] / Ntot
                nat.call("vmcpde_row_moments_sq", nat.ptr(JT), ldj, P, n, nat.ptr(e), None, nat.ptr(Em.reshape(1).contiguous()), nat.ptr(w["rowsum"]),
                         nat.ptr(w["rowdot"]), nat.ptr(w["rowsq"]), nat.stream())       # one pass: sum J, sum J (E - Em), sum J^2
                if world > 1:
                    mom = torch.stack([w["rowsum"], w["rowdot"], w["rowsq"]])
                    mpi.allreduce_sum_(mom)
                    rowsum, rowdot, rowsq = mom[0], mom[1], mom[2]
                else:
                    rowsum, rowdot, rowsq = w["rowsum"], w["rowdot"], w["rowsq"]
                mean = rowsum / Ntot
                F = rowdot / Ntot                                    # sum_n (J - m)(E - Em) = sum_n J (E - Em)  (Em is the global mean)
                diagS = rowsq / Ntot - mean * mean
                if self.frozenParameters is not None:
                    raise NotImplementedError("matrixFree with frozenParameters")
                dmax = diagS.max()
                floor = float(self.svdTol) * dmax
                D = diagS * float(self.diagonalShift) + floor        # (Cholesky-type solve only)
                scal = torch.stack([Em, Eabs, E2 - Em * Em, E2, torch.zeros_like(Em), dmax, floor, torch.zeros_like(Em)])
                cand_off = nat.lib().vmcpde_lowrank_candidates_byte_offset(P)
                cand_ptr = nat.C.c_void_p(w["ws"].data_ptr() + cand_off)                       # the step's int32[32] candidate rows
                cand_view = w["ws"].view(torch.int32)[cand_off // 4:cand_off // 4 + 32]
                wsb, gcb = w["ws"].numel() * 8, w["gc_ws"].numel() * 8
                plan = self._mf_plan
                if plan is not None and plan["event"] is not None:
                    plan["event"].synchronize()
                    if not int(plan["host"][1]) or int(plan["host"][0]) > plan["rmax"] - 32:
                        plan = self._mf_plan = None
                    else:
                        used = self._steps_used(plan["head"])
                        plan["nsteps"] = used + max(3, used // 3)
                        plan["prev_rank"] = int(plan["host"][0])
                        plan["prev_passes"] = int(plan["head"].view(torch.int32)[8])      # block steps that needed their own pass over J
                rmax = plan["rmax"] if plan is not None else cap
                skip_ptr = nat.C.c_void_p(w["ws"].data_ptr() + nat.lib().vmcpde_lowrank_skip_flag_byte_offset())

                def block_step():
                    nat.call("vmcpde_lowrank_step", None, P, tol, min(rmax, P), nat.ptr(w["LT"]), ldl, nat.ptr(w["piv"]), nat.ptr(w["state"]), nat.ptr(w["ws"]),
                             wsb, nat.stream())
                    if world == 1:
                        nat.call("vmcpde_gram_columns", nat.ptr(JT), ldj, P, n, cand_ptr, nat.ptr(mean), Ntot, nat.ptr(w["Scol"]), ldl,
                                 skip_ptr, nat.ptr(w["gc_ws"]), gcb, nat.stream())      # (skipped when the step is done or served ahead)
                    else:        # raw local sums, one 1 MB all-reduce, then 1/N and the mean term (every rank chose the same candidates)
                        nat.call("vmcpde_gram_columns", nat.ptr(JT), ldj, P, n, cand_ptr, None, 1.0, nat.ptr(w["Scol"]), ldl, None,
                                 nat.ptr(w["gc_ws"]), gcb, nat.stream())
                        mpi.allreduce_sum_(w["Scol"])
                        rows = cand_view.long()
                        w["Scol"][:, :P] = w["Scol"][:, :P] / Ntot - torch.outer(mean[rows], mean)
                    nat.call("vmcpde_lowrank_step", nat.ptr(w["Scol"]), P, tol, min(rmax, P), nat.ptr(w["LT"]), ldl, nat.ptr(w["piv"]), nat.ptr(w["state"]),
                             nat.ptr(w["ws"]), wsb, nat.stream())

                if plan is None:
                    w["LT"].zero_()          # (planning: the pad of the plan made below may reach past this call's rank cap)
                else:
                    w["LT"][:rmax].zero_()
                self._arm_hints(w, P)
                hints_on = bool(self.pivotHints) and os.environ.get("VMCPDE_PIVOT_HINTS", "1") != "0" and os.environ.get("VMCPDE_HINT_COLUMNS", "1") != "0"
                ahead = 0
                if plan is not None and hints_on and plan.get("prev_rank", plan["rank"]) > 0:
                    # the columns of S at the previous call's pivots -- this call's preferred candidates -- in ONE pass over J: gather of
                    # those rows, a GEMM batched over sample chunks (one library GEMM with K = n runs on a handful of workgroups), the
                    # mean term; the block steps whose candidates all come from the list then need no pass of their own
                    ahead = min(_round_up(plan.get("prev_rank", plan["rank"]), 32), rmax, P)
                    idx = w["piv"][:ahead].long()               # (read before the factorisation overwrites it: stream order)
                    JH = Jv[idx]                                 # ahead x n, contiguous
                    ckn = next((c for c in (4096, 2048, 1024, 128) if n % c == 0), 0)
                    if ckn:
                        SH = torch.bmm(JH.view(ahead, n // ckn, ckn).permute(1, 0, 2), Jv.view(P, n // ckn, ckn).permute(1, 2, 0)).sum(0)
                    else:
                        SH = JH @ Jv.T
                    if world > 1:
                        mpi.allreduce_sum_(SH)
                    if w.get("Sbig") is None or w["Sbig"].shape[0] < ahead:
                        w["Sbig"] = torch.zeros((w["LT"].shape[0], ldl), dtype=global_defs.tReal, device=device)
                    torch.addcmul(SH / Ntot, mean[idx][:, None], mean[None, :], value=-1.0, out=w["Sbig"][:ahead, :P])
                nat.call("vmcpde_lowrank_begin", nat.ptr(diagS), P, nat.ptr(w["state"]), nat.ptr(w["ws"]), wsb, nat.stream())
                if ahead:
                    nat.call("vmcpde_lowrank_columns", nat.ptr(w["Sbig"]), ahead, P, nat.ptr(w["ws"]), wsb, nat.stream())
                if plan is None:                                     # planning: run until the criterion is met, reading the state every 4 steps
                    steps, done = 0, False
                    while steps < cap:
                        for _ in range(4):
                            block_step()
                        steps += 4
                        nat.call("vmcpde_lowrank_step", None, P, tol, min(rmax, P), nat.ptr(w["LT"]), ldl, nat.ptr(w["piv"]), nat.ptr(w["state"]), nat.ptr(w["ws"]),
                                 wsb, nat.stream())           # (a selection settles `state`)
                        w["state_host"].copy_(w["state"], non_blocking=True)
                        torch.cuda.current_stream(device).synchronize()
                        if int(w["state_host"][1]) or int(w["state_host"][0]) >= cap:
                            done = bool(int(w["state_host"][1]))
                            break
                    r = int(w["state_host"][0])
                    if not done or r < 1:
                        raise FloatingPointError(f"matrixFree: S is not rank deficient within the cap (rank >= {r} of {P}); use the assembled path")
                    rmax = _round_up(r + 40, 128)
                    self._mf_plan = dict(rmax=rmax, nsteps=steps + 3, rank=r, event=None, host=torch.zeros(2, dtype=torch.int32).pin_memory(),
                                         head=torch.zeros(5, dtype=global_defs.tReal).pin_memory())
                    mode = "planning"
                else:
                    for _ in range(plan["nsteps"]):
                        block_step()
                    nat.call("vmcpde_lowrank_step", None, P, tol, min(rmax, P), nat.ptr(w["LT"]), ldl, nat.ptr(w["piv"]), nat.ptr(w["state"]), nat.ptr(w["ws"]), wsb,
                             nat.stream())
                    plan["host"].copy_(w["state"], non_blocking=True)
                    plan["head"].copy_(w["ws"][:5], non_blocking=True)
                    if plan["event"] is None:
                        plan["event"] = torch.cuda.Event()
                    plan["event"].record()
                    mode = "static"
            with self._Section(self, "allreduce"):
                pass
            self._replicated = self._Section(self, "replicated")
            self._replicated.__enter__()

            need_mv = nat.lib().vmcpde_jt_matvec_workspace_bytes(P, n)
            if w.get("mv_ws") is None or w["mv_ws"].numel() * 8 < need_mv or w["mv_z"].numel() < n:
                w["mv_ws"] = torch.empty(need_mv // 8, dtype=global_defs.tReal, device=device)
                w["mv_z"] = torch.empty(n, dtype=global_defs.tReal, device=device)

            def s0(v):                                               # J^T (J v) / N - m (m.v): two passes over J (+ a P-vector all-reduce)
                y = torch.empty(P, dtype=global_defs.tReal, device=device)
                nat.call("vmcpde_jt_matvec", nat.ptr(JT), ldj, P, n, nat.ptr(v.contiguous()), nat.ptr(w["mv_z"]), nat.ptr(y), nat.ptr(w["mv_ws"]),
                         w["mv_ws"].numel() * 8, nat.stream())
                if world > 1:
                    mpi.allreduce_sum_(y)
                return y / Ntot - mean * (mean @ v)

            if spectral:
                # the reference's spectral solve on the same factor: exact rank (one host read), Householder basis Q1 of range(L),
                # Q1^T S Q1 = (J Q1)^T (J Q1) / N - (Q1^T m)(Q1^T m)^T from ONE more pass over J, Ritz pairs, V = [Q2 | Q1 W]
                def read_state():
                    w["state_host"].copy_(w["state"], non_blocking=True)
                    torch.cuda.current_stream(device).synchronize()
                    return int(w["state_host"][0]), int(w["state_host"][1])

                r, fin = read_state()
                extra = 0
                while not fin and r < rmax and extra < 16:            # (the plan's steps did not suffice for these samples: go on, the host is here anyway)
                    block_step()
                    block_step()
                    nat.call("vmcpde_lowrank_step", None, P, tol, min(rmax, P), nat.ptr(w["LT"]), ldl, nat.ptr(w["piv"]), nat.ptr(w["state"]), nat.ptr(w["ws"]), wsb,
                             nat.stream())
                    extra += 2
                    r, fin = read_state()
                if extra:
                    self._mf_plan["nsteps"] += extra
                if not fin or r < 1:
                    self._mf_plan = None
                    raise StaticPlanOutdated(f"matrixFree: the factorisation did not finish within its plan (rank {r}); the next call plans anew")
                self._lowrank_last_rank = r
                Y, Xn, Q1 = self._lowrank_range_basis(w, P, r, device)
                Z = Q1.T @ Jv                                        # r x n
                ckn = 2048 if n % 2048 == 0 else (128 if n % 128 == 0 else 0)
                if ckn:                                              # (K = n, result r x r: batched over sample chunks to fill the chip)
                    Z3 = Z.view(r, n // ckn, ckn).permute(1, 0, 2)
                    Tm = torch.bmm(Z3, Z3.transpose(1, 2)).sum(0)
                else:
                    Tm = Z @ Z.T
                if world > 1:
                    mpi.allreduce_sum_(Tm)
                q = Q1.T @ mean
                Tm = Tm / Ntot - torch.outer(q, q)
                wv, W = torch.linalg.eigh(0.5 * (Tm + Tm.T))
                self.ev, self.V = self._lowrank_assemble(Y, Xn, Q1 @ W, wv, r, P, device)
                self.spectralInfo = dict(engine="matrix-free", mode=mode, rank=r, block_steps=self._mf_plan["nsteps"], columns_ahead=ahead,
                                         own_passes_last_call=self._mf_plan.get("prev_passes"))
                Fc = F.contiguous()
                self.VtF = torch.empty(P, dtype=global_defs.tReal, device=device)
                nat.call("vmcpde_gemv", nat.ptr(self.V), P, P, P, 1, nat.ptr(Fc), nat.ptr(self.VtF), nat.stream())
                coef = torch.empty(P, dtype=global_defs.tReal, device=device)
                nat.call("vmcpde_pinv_coeffs", nat.ptr(self.ev), nat.ptr(self.VtF), None, P, float(self.svdTol), float(self.snrTol), nat.ptr(coef),
                         nat.stream())
                update = torch.empty(P, dtype=global_defs.tReal, device=device)
                nat.call("vmcpde_gemv", nat.ptr(self.V), P, P, P, 0, nat.ptr(coef), nat.ptr(update), nat.stream())
                S0u = s0(update)
                out = torch.empty(4, dtype=global_defs.tReal, device=device)
                nat.call("vmcpde_tdvp_diagnostics", nat.ptr(update), nat.ptr(S0u), nat.ptr(Fc), nat.ptr(diagS.contiguous()), P, 0.0, None, nat.ptr(scal),
                         None, nat.ptr(out), nat.stream())
                self._replicated.__exit__(None, None, None)
                self._record_guard(out, False)
                self.ElocMean, self.ElocMeanAbs, self.ElocVar = scal[0], scal[1], scal[2]
                self._scal, self._floor = scal, None
                self.F0, self.S0 = F, None
                self._S0_lazy = True
                self._lazy = dict(JT=JT, ldj=ldj, P=P, n=n, Ntot=Ntot, e=e, logp=logProbs.reshape(-1), b=None, mean=mean, Emean=scal[0],
                                  chol_border=False, owner=psi, gen=psi.workspace_generation(n))
                self._SExp = self._rhoVar = self._snr = self._S = None
                self.cholInfo = None
                return update, out[0], out[1]

            rmax = self._mf_plan["rmax"]
            if w.get("G") is None or w["G"].shape[0] != rmax:
                need = nat.lib().vmcpde_chol_workspace_bytes(rmax)
                w["G"] = torch.zeros((rmax, rmax), dtype=global_defs.tReal, device=device)
                w["G_ws"] = torch.empty(need // 8, dtype=global_defs.tReal, device=device)
                w["G_info"] = torch.zeros(1, dtype=torch.int32, device=device)
                w["t2"] = torch.empty(rmax, dtype=global_defs.tReal, device=device)
            G, gws, t2 = w["G"], w["G_ws"], w["t2"]
            need = gws.numel() * 8
            ck = 128 if ldl % 128 == 0 else 64
            if w.get("Dpad") is None:
                w["Dpad"] = torch.ones(ldl, dtype=global_defs.tReal, device=device)
            w["Dpad"][:P] = D
            Lfull = w["LT"][:rmax]
            LTr = Lfull[:, :P]
            A3 = (Lfull / w["Dpad"]).view(rmax, ldl // ck, ck).permute(1, 0, 2)
            B3 = Lfull.view(rmax, ldl // ck, ck).permute(1, 2, 0)
            torch.sum(torch.bmm(A3, B3), dim=0, out=G)
            G.diagonal().add_(1.0)
            nat.call("vmcpde_cholesky", nat.ptr(G), rmax, rmax, nat.ptr(w["G_info"]), nat.ptr(gws), need, nat.stream())

            def minv(v):
                y = v / D
                t1 = LTr @ y
                nat.call("vmcpde_chol_solve", nat.ptr(G), rmax, rmax, nat.ptr(t1), nat.ptr(t2), nat.ptr(gws), need, nat.stream())
                return torch.addcdiv(y, LTr.T @ t2, D, value=-1.0)

            u = minv(F)
            u = u + minv(F - s0(u) - D * u)
            update = u.contiguous()
            S0u = s0(update)
            out = torch.empty(4, dtype=global_defs.tReal, device=device)
            info = torch.zeros(1, dtype=torch.int32, device=device)
            floor_t = floor.reshape(1).contiguous()
            nat.call("vmcpde_tdvp_diagnostics", nat.ptr(update), nat.ptr(S0u), nat.ptr(F), nat.ptr(diagS.contiguous()), P,
                     float(self.diagonalShift) if self.diagonalShift > 1e-10 else 0.0, nat.ptr(floor_t), nat.ptr(scal), nat.ptr(info), nat.ptr(out),
                     nat.stream())
            self._replicated.__exit__(None, None, None)
            self._record_guard(out, True)
        self.ElocMean, self.ElocMeanAbs, self.ElocVar = scal[0], scal[1], scal[2]
        self._scal, self._floor = scal, floor_t
        self.F0, self.S0 = F, None
        self._S0_lazy = True
        self._lazy = dict(JT=JT, ldj=ldj, P=P, n=n, Ntot=Ntot, e=e, logp=logProbs.reshape(-1), b=None, mean=mean, Emean=scal[0], chol_border=False,
                          owner=psi, gen=psi.workspace_generation(n))
        self._SExp = self._rhoVar = self._snr = self._S = None
        self.ev = self.V = self.VtF = None
        self.cholInfo = info
        self.choleskyInfo = dict(engine="matrix-free", mode=mode, rmax=rmax, block_steps=self._mf_plan["nsteps"], planned_rank=self._mf_plan["rank"],
                                 columns_ahead=ahead, own_passes_last_call=self._mf_plan.get("prev_passes"))
        return update, out[0], out[1]

    # ---- tdvp.py:73-94 -------------------------------------------------------------------------------------------
    def solve(self, Eloc, gradients, logProbs, psi=None):
        if self.matrixFree:
            return self._solve_matrix_free(Eloc, gradients, logProbs, psi)
        F = self.get_tdvp_equation(Eloc, gradients, logProbs, psi=psi)
        b = self._lazy["b"]
        P = F.numel()
        device = F.device
        update = torch.empty(P, dtype=global_defs.tReal, device=device)
        with torch.cuda.device(device):
            lowrank_done = False
            if self._solver() == "cholesky":
                try_lr = self._lowrank_tol() is not None and (self.choleskyEngine == "lowrank" or
                                                              (self.choleskyEngine == "auto" and P >= LOWRANK_CHOL_MIN_P and self._chol_backoff == 0))
                if self._lowrank_tol() is None:
                    self._chol_static = None
                if self.choleskyEngine == "auto" and self._chol_backoff > 0:
                    self._chol_backoff -= 1
                self.choleskyInfo = dict(engine="dense")
                if try_lr:
                    lowrank_done = self._solve_lowrank(b, F, P, update)
                    if not lowrank_done:
                        self.choleskyInfo = dict(self.choleskyInfo, engine="dense", tried="lowrank")
                        if self.choleskyEngine == "auto":
                            self._chol_backoff = LOWRANK_RETRY_AFTER
            if lowrank_done:
                self.ev = self.V = self.VtF = None
                self.cholInfo = b["info"]
            elif self._solver() == "cholesky":
                A, Ppad = b["A"], b["Ppad"]
                bordered = P < Ppad          # a spare padded row: factor [[A, F], [F^T, beta]] -> its last row is y = L^-1 F
                Pf = P + 1 if bordered else P
                need = nat.lib().vmcpde_chol_workspace_bytes(Pf)
                if b["chol_ws"] is None or b["chol_ws"].numel() * 8 < need:
                    b["chol_ws"] = torch.empty(need // 8, dtype=global_defs.tReal, device=device)
                # (A = S + shift diag(S) + floor I, and the border row [F, beta] when `bordered`, were written by get_tdvp_equation)
                nat.call("vmcpde_cholesky", nat.ptr(A), Ppad, Pf, nat.ptr(b["info"]), nat.ptr(b["chol_ws"]), need, nat.stream())
                if bordered:     # the forward substitution rode along with the factorisation (32 dependent launches less)
                    nat.call("vmcpde_chol_backsolve", nat.ptr(A), Ppad, P, nat.ptr(A[P]), nat.ptr(update), nat.ptr(b["chol_ws"]), need,
                             nat.stream())
                else:
                    nat.call("vmcpde_chol_solve", nat.ptr(A), Ppad, P, nat.ptr(F), nat.ptr(update), nat.ptr(b["chol_ws"]), need,
                             nat.stream())
                self.ev = self.V = self.VtF = None
                self.cholInfo = b["info"]
            else:
                self.transform_to_eigenbasis(self.S, F)
                coef = torch.empty(P, dtype=global_defs.tReal, device=device)
                snr = self.snr.contiguous() if self.useSNR else None                                   # tdvp.py:87-89
                nat.call("vmcpde_pinv_coeffs", nat.ptr(self.ev), nat.ptr(self.VtF), nat.ptr(snr), P, float(self.svdTol),
                         float(self.snrTol), nat.ptr(coef), nat.stream())                              # tdvp.py:82-85
                nat.call("vmcpde_gemv", nat.ptr(self.V), P, P, P, 0, nat.ptr(coef), nat.ptr(update), nat.stream())   # tdvp.py:91
            # residual and TDVP error (tdvp.py:93-94) from one S0 u: the shifted S adds (diagonalShift diag(S0) + floor) u
            S0u = torch.empty(P, dtype=global_defs.tReal, device=device)
            out = torch.empty(4, dtype=global_defs.tReal, device=device)
            chol = self._solver() == "cholesky"
            nat.call("vmcpde_gemv", nat.ptr(b["S"]), b["Ppad"], P, P, 0, nat.ptr(update), nat.ptr(S0u), nat.stream())
            nat.call("vmcpde_tdvp_diagnostics", nat.ptr(update), nat.ptr(S0u), nat.ptr(b["F"]), nat.ptr(b["diagS"]), P,
                     float(self.diagonalShift) if self.diagonalShift > 1e-10 else 0.0,
                     nat.ptr(self._floor) if (chol and self._floor is not None) else None, nat.ptr(self._scal),
                     nat.ptr(b["info"]) if chol else None, nat.ptr(out), nat.stream())
            self._replicated.__exit__(None, None, None)
            self._record_guard(out, chol)
        return update, out[0], out[1]

    # ---- state carried from call to call (not in the reference, whose calls are independent) ---------------------
    def state_dict(self):
        """What a call inherits from the previous one: the global means used as the shift of the one-pass moments and the Gram
        (centering / gramShift "auto").  With it a resumed run repeats the uninterrupted one bit for bit (examples/main.py) on the
        dense solve paths.  The low-rank routes (P >= LOWRANK_CHOL_MIN_P / LOWRANK_MIN_P, matrixFree) also inherit a plan and the
        previous pivots, which are NOT part of this record: a resumed run starts with a planning call and no hint list, and differs
        from the uninterrupted one at the level of the solve's accuracy (1e-10 |F|) from there on."""
        if self._ratio_event is not None:
            self._ratio_event.synchronize()
            self._uncentred_ok = bool(self._ratio_host[0] < 1e3)
        return {"c_prev": None if self._c_prev is None else self._c_prev.detach().cpu().clone(),
                "cE_prev": None if self._c_prev is None else self._cE_prev.detach().cpu().clone(),
                "uncentred_ok": bool(self._uncentred_ok)}

    def load_state_dict(self, st, device=None):
        device = global_defs.my_device() if device is None else device
        self._c_prev = None if st.get("c_prev") is None else st["c_prev"].to(device)
        self._cE_prev = None if st.get("cE_prev") is None else st["cE_prev"].to(device)
        self._uncentred_ok = bool(st.get("uncentred_ok", False))
        self._ratio_host = self._ratio_event = None      # the restored flag stands until the next call's own check lands

    # ---- tdvp.py:143-162 -----------------------------------------------------------------------------------------
    def _observables(self, sampleConfigs, logProbs, Eloc):
        """mean, covariance (1/n), entropy, central moments 3-6 of the samples and the largest Eloc: `vmcpde_observables` (two
        launches) for fp64 samples on the device, the reference's array expressions otherwise."""
        d = sampleConfigs.shape[-1]
        x = sampleConfigs.reshape(-1, d)
        info = {}
        if x.is_cuda and x.dtype == global_defs.tReal and d <= 16 and x.shape[0] >= 1:
            x = x.contiguous()
            lp, el = logProbs.reshape(-1).contiguous(), Eloc.reshape(-1).contiguous()
            ws = getattr(self, "_obs_ws", None)
            if ws is None or ws.device != x.device:
                ws = self._obs_ws = torch.zeros(nat.lib().vmcpde_observables_workspace_bytes() // 8, dtype=global_defs.tReal, device=x.device)
            out = torch.empty(d * d + 5 * d + 2, dtype=global_defs.tReal, device=x.device)
            with torch.cuda.device(x.device):
                # (Eloc keeps the nSamplesTDVP entries of the solve when the samples were re-drawn at nSamplesObs, tdvp.py:130-134,150)
                nat.call("vmcpde_observables", nat.ptr(x), x.shape[0], d, nat.ptr(lp), nat.ptr(el), el.numel(), nat.ptr(out), nat.ptr(ws),
                         ws.numel() * 8, nat.stream())
            info["x1"] = out[:d]
            info["covar"] = out[d:d + d * d].reshape(d, d)
            info["entropy"] = out[d * d + 5 * d]
            for k, m in enumerate([3, 4, 5, 6]):
                info[f"x{m}"] = out[d + d * d + k * d:d + d * d + (k + 1) * d]
            info["max_grad"] = out[d * d + 5 * d + 1]
            return info
        mean = sampleConfigs.mean(dim=(0, 1), keepdim=True)
        info["x1"] = mean[0, 0]
        X = sampleConfigs[0] - mean[0]
        # (a mean of outer products, not `X.T @ X`: for a d x N by N x d product rocBLAS runs one 128 x 128 macro tile over all N
        # samples -- 6.9 ms at N = 65 536, a third of a TDVP step; tools/diag/observables_prof.sh)
        info["covar"] = (X[:, :, None] * X[:, None, :]).mean(dim=0)
        info["entropy"] = -logProbs.mean()
        for m in [3, 4, 5, 6]:
            info[f"x{m}"] = ((sampleConfigs - mean) ** m).mean(dim=(0, 1))
        info["max_grad"] = Eloc.max()
        return info

    def _ball_samples(self, psi, n, dim):
        """Uniform points in the unit ball (1,n,dim) for the small-ball integrals (tdvp.py:152-157).  The reference draws them from
        `sampler.key`, which every sampler call advances (sampler.py:73): fresh points per call, here from the sampler's key and
        call count.  Departure, on purpose: the reference passes the SAME key to `normal` and `uniform` (tdvp.py:154-155), which
        correlates direction and radius of a point through the shared threefry bits; here the two are independent draws, so the
        points are exactly uniform in the ball.  Tests inject points by replacing this method on the instance."""
        g = torch.Generator(device=psi.device).manual_seed(int(psi.sampler.key) + 104729 + 7919 * int(getattr(psi.sampler, "_calls", 0)))
        samples = torch.randn((1, n, dim), generator=g, dtype=global_defs.tReal, device=psi.device)
        radius = torch.rand((1, n), generator=g, dtype=global_defs.tReal, device=psi.device)
        return samples / torch.linalg.norm(samples, dim=-1, keepdim=True) * radius[..., None] ** (1 / dim)

    # ---- tdvp.py:96-164 ------------------------------------------------------------------------------------------
    def __call__(self, netParameters, t, psi, evolutionEq, **rhsArgs):
        nSamplesTDVP = rhsArgs["nSamplesTDVP"]
        nSamplesObs = rhsArgs["nSamplesObs"]
        mpi.globNumSamples = nSamplesTDVP                                     # tdvp.py:99
        psi.set_parameters(netParameters)                                     # tdvp.py:101
        timings = rhsArgs["timings"]

        def start_timing(name):
            if timings is not None:
                torch.cuda.synchronize(psi.device)
                timings.start_timing(name)

        def stop_timing(name):
            if timings is not None:
                torch.cuda.synchronize(psi.device)
                timings.stop_timing(name)

        with self._Section(self, "sample+flow"):
            start_timing("sampling")
            sampleConfigs, logProbs = psi.sample(numSamples=nSamplesTDVP)         # tdvp.py:117
            stop_timing("sampling")

            start_timing("compute Eloc")
            Eloc, sampleGradients, logProbs = evolutionEq(psi, sampleConfigs, t)  # tdvp.py:122
            stop_timing("compute Eloc")

        start_timing("solve TDVP eqn.")
        update, self.solverResidual, self.tdvp_error = self.solve(Eloc, sampleGradients, logProbs, psi=psi)   # tdvp.py:127
        stop_timing("solve TDVP eqn.")

        if nSamplesObs > nSamplesTDVP:                                        # tdvp.py:130-134
            start_timing("sampling observables")
            sampleConfigs, logProbs = psi.sample(numSamples=nSamplesObs)
            stop_timing("sampling observables")

        if not rhsArgs.get("observables", True):
            return update, {}

        try:
            self._check_guard("this step")      # (the host synchronises for the NaN check below anyway)
        except StaticPlanOutdated:              # a synchronous call repairs itself: the same samples through a planning call
            update, self.solverResidual, self.tdvp_error = self.solve(Eloc, sampleGradients, logProbs, psi=psi)
            self._check_guard("this step")
        if bool(torch.isnan(update).any()):                                   # tdvp.py:136-141
            print(self.S0)
            print(self.F0)
            raise FloatingPointError("nan encountered in the TDVP update")

        info = self._observables(sampleConfigs, logProbs, Eloc)               # tdvp.py:143-150

        # tdvp.py:152-162.  The reference evaluates them on every call and ignores the `integrals` keyword its driver passes
        # (main.py:111,161 -- the published records hold them although `comp_integrals = False`); here they are on by default and
        # `integrals=False` (or None) switches the three extra flow evaluations off.
        if rhsArgs.get("integrals", True) not in (None, False):
            dim = sampleConfigs.shape[-1]
            samples = self._ball_samples(psi, nSamplesObs, dim)
            for lim_normal in [1, 0.5, 0.1]:
                T = 10
                lim = lim_normal * math.sqrt(T)
                sphere_volume = math.pi ** (dim / 2) / math.gamma(dim / 2 + 1) * lim ** dim
                info[f"integral_{lim_normal}sigma"] = torch.mean(torch.exp(psi(lim * samples))) * sphere_volume
        return update, info
