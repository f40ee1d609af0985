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
  * from 1024 / 4096 parameters on both solvers go through a pivoted low-rank factor of S when S is numerically rank deficient
    (vmc_pde_amd/lowrank.py); the dense factorisations remain the fallback;
  * `SExp`, `rhoVar`, `snr` are evaluated lazily from the retained Jacobian (weighted Gram), since only AdaptiveHeun
    (stepper.py:71) and useSNR / the driver's logging (main.py:187) read them.

This file is the product path only.  The two opt-in experiments (an S-build on the int8 matrix pipe, a step without an assembled S)
live in vmc_pde_amd/experimental/ as a subclass; the hooks they override are `_gram_launch`, `_lowrank_tol`, `_drop_plans`, `solve`.
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
from .gram_operator import GramOperator, _VecTimesGram  # noqa: F401
from .lowrank import (LowRankRoutes, StaticPlanOutdated, _long_k_product, _round_up,  # noqa: F401
                      LOWRANK_MIN_P, LOWRANK_RANK_CAP, LOWRANK_RANK_DIVISOR, LOWRANK_TOL, LOWRANK_TRACE_BOUND, LOWRANK_SVDTOL_FRACTION,
                      LOWRANK_CHOL_MIN_P, LOWRANK_SOLVE_RHO, LOWRANK_RETRY_AFTER, LOWRANK_GATE_RETRY_AFTER)


F32_BLOCK_MIN_SHIFT = 1e-4   # smallest diagonalShift the fp32-block Gram is allowed with (see TDVP._gram)
_BORDER_PIVOT = 1e200    # corner of the bordered matrix: any value above F^T A^-1 F keeps the last pivot positive


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


DENSE_EIGH_WARN_P = 2048    # from this many parameters on a call that ends in the dense syevd says so once (its cost is O(P^3), replicated per rank)
DEVICE_EIGH_MIN_P = 640     # diagonalizeOnDevice=None: rocSOLVER's syevd overtakes thread-capped host LAPACK between P = 411 and 1024


@dataclass
class TDVP(LowRankRoutes):
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
    coreEigensolver: str = "auto" # the r x r symmetric eigenproblem inside the rank-revealing spectral engine.  "auto" (default) = "tridiag" while the
                                  #   rank fits its kernel (r <= 256): own Householder tridiagonalisation in ONE workgroup with the lower
                                  #   triangle in registers (csrc/tridiag.hip), rocSOLVER's divide & conquer on the tridiagonal matrix, own
                                  #   back-transformation -- the library's syevd spends 2.9 of its 4.3 ms at r = 206 in the launch chain of its
                                  #   tridiagonalisation; "library" -- torch.linalg.eigh =
                                  #   rocSOLVER's syevd (a chain of some hundred launches: 4.35 of the engine's 6.8 ms at the headline size);
                                  #   "jacobi" -- the own kernel, blocked one-sided Jacobi on the triangular factor R1 of L = Q1 R1
                                  #   (csrc/jacobi.hip).  Correct and tested, but SLOWER (11 ms at the headline size: every rotation moves two
                                  #   whole columns through the LDS for three flops per element) -- kept as a measured
                                  #   negative result; VMCPDE_CORE_EIGH=jacobi for an A/B.
    choleskyEngine: str = "auto"  # how solver="cholesky" solves (S + diagonalShift diag(S) + floor I) u = F.  "dense": the blocked factorisation of
                                  #   the full matrix (csrc/solve.hip: 2.0 ms at P = 4213, 36 ms at 16 181).  "lowrank": S = L L^T + R by the
                                  #   pivoted Cholesky of csrc/eig.hip (r = 190 columns of 4213), the diagonal part D through the Woodbury
                                  #   identity -- a Cholesky of the r x r matrix I + L^T D^-1 L -- and one or two steps of iterative refinement
                                  #   against the TRUE matrix S + D, which remove what R and the identity's cancellation left (the solution
                                  #   of the same system to ~1e-14).  "auto": lowrank from LOWRANK_CHOL_MIN_P parameters on (where the O(P^3)
                                  #   factorisation costs more than the r dependent block steps) while the rank stays under the cap and
                                  #   trace(R) under 1e-4 min D; dense otherwise (and for the next LOWRANK_RETRY_AFTER calls).
    frozenPivots: bool = True     # Cholesky route on a static plan: up to FROZEN_REFRESH_AFTER consecutive calls factor S on the PREVIOUS factorisation's pivot set
                                  # without searching (gather, one tall Cholesky, transpose: 0.2 ms instead of 0.57 ms at the headline size); the remainder R
                                  # is measured in every such call and a set that no longer spans S sends the next call back to the pivoted factorisation.
                                  # Every call still refines against the true matrix and is guarded by its residual.  VMCPDE_FROZEN_PIVOTS=0 for an A/B.
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
    overlapAllreduce: bool = False  # sharded samples, steady-state Gram: the S-build runs as groups of tile rows (vmcpde_gram_rows) and every
                                  # finished group -- one contiguous piece of the packed buffer -- is all-reduced on a side stream while
                                  # the next is being formed; finalize waits for the last.  Same bits as the single call.  Off by default:
                                  # it trades a partly filled round of the Gram's schedule per group for the hidden collective and has never met more
                                  # than one GPU (VMCPDE_OVERLAP_ALLREDUCE=1; VMCPDE_OVERLAP_FRACTIONS="0.8" -- the default: two groups cut at 80 % of the tiles --
                                  # or VMCPDE_OVERLAP_GROUPS=G equal groups, for an A/B on a node)
    f32BlockGram: bool = False    # with an fp32 Jacobian: fp32 MFMA sums over blocks of <= 128 samples, fp64 across blocks (2x MFMA rate)
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
        self.choleskyInfo = None
        self._lowrank_backoff = 0
        self._lowrank_last_rank = 0
        self.spectralInfo = None
        if self.solver == "cholesky" and self.useSNR:
            # tdvp.py:87-89: the SNR soft cut-off lives in the eigenbasis of S; a Cholesky factor has none
            raise ValueError("useSNR=True needs the spectral solver (solver='eigh' or 'auto'); the Cholesky path has no eigenbasis")
        self._guard_host = self._guard_event = None
        self._guard_pending = self._guard_chol = False
        self.cholInfo = None
        self._buf = {}
        self._c_prev = None
        self._lazy = None
        self._SExp = self._rhoVar = self._snr = self._S = None
        self.S0 = None
        self._V = self._V_lazy = None
        self.ev = self.V = self.VtF = None
        self._ratio_host = self._ratio_event = None
        self._ratio_queued = self._static_queued = None
        self._ratio_prev = None
        self.allreduceInfo = None
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
            self._gram_launch(b, JT, ldj, f32, P, n, c, w, out)

    def _gram_and_allreduce(self, b, JT, ldj, Pg, n, c_gram):
        """The S-build of this rank's shard into b["packed"] and the step's ONE collective over b["pack"] = [tiles | sum O | sum E~O |
        Eloc sums | n] (mpi_wrapper.py:129-163) -- back to back, or (overlapAllreduce) interleaved by groups of tile rows."""
        overlap = (self.overlapAllreduce or os.environ.get("VMCPDE_OVERLAP_ALLREDUCE", "0") == "1") and c_gram is None and \
            JT.dtype == global_defs.tReal and not getattr(self, "int8Gram", False) and not self.f32BlockGram
        nb = (Pg + nat.GRAM_TILE - 1) // nat.GRAM_TILE
        ngroups = min(max(int(os.environ.get("VMCPDE_OVERLAP_GROUPS", "2")), 1), nb)
        if not overlap or ngroups < 2 or nb < 2:
            self._gram(b, JT, ldj, Pg, n, c_gram, None, b["packed"])
            with self._Section(self, "allreduce"):
                mpi.allreduce_sum_(b["pack"])        # the one collective of the step (no-op on a single rank)
            self.allreduceInfo = dict(overlapped=False)
            return
        need = nat.lib().vmcpde_gram_workspace_bytes(Pg, n)
        if b["gram_ws"] is None or b["gram_ws"].numel() * 8 < need:
            b["gram_ws"] = torch.empty(max(need // 8, 1), dtype=global_defs.tReal, device=JT.device)
        # tile rows dealt into groups (row r holds r + 1 tiles): about equal tile counts, or -- VMCPDE_OVERLAP_FRACTIONS="0.5,0.8" -- cut at
        # those cumulative shares of the tiles (a small LAST group leaves little of the collective exposed behind the S-build); the last
        # group also carries the buffer's tail
        tri = lambda r: r * (r + 1) // 2
        # (default: ONE cut at 0.8 -- measured on one GPU at the C4 shard, where nothing is hidden: the Gram 8.88 ms as one launch, 9.06
        # with this cut, 9.24 with cuts at 0.5 / 0.8, 9.59 with four equal groups)
        frac_env = os.environ.get("VMCPDE_OVERLAP_FRACTIONS", "" if "VMCPDE_OVERLAP_GROUPS" in os.environ else "0.8")
        cuts = [float(v) for v in frac_env.split(",") if v.strip()]
        if cuts:
            ngroups = min(len(cuts) + 1, nb)
            cuts = sorted(min(max(v, 0.0), 1.0) for v in cuts)[:ngroups - 1]
        else:
            cuts = [g / ngroups for g in range(1, ngroups)]
        bounds, r = [0], 0
        for g in range(1, ngroups):
            while r < nb and tri(r) < cuts[g - 1] * tri(nb):
                r += 1
            bounds.append(min(max(r, bounds[-1] + 1), nb - (ngroups - g)))
        bounds.append(nb)
        main = torch.cuda.current_stream(JT.device)
        if getattr(self, "_side_stream", None) is None or self._side_stream.device != JT.device:
            self._side_stream = torch.cuda.Stream(device=JT.device)
            self._group_streams = []
        # group launches on streams of their own (VMCPDE_OVERLAP_STREAMS=0: all on the caller's stream, where every group's last, partly
        # filled round of workgroups is a bubble): the next group's workgroups fill the CUs the previous one frees; the first
        # groups get the higher priority so that they still finish first
        concurrent = os.environ.get("VMCPDE_OVERLAP_STREAMS", "1") != "0"
        while concurrent and len(self._group_streams) < ngroups:
            self._group_streams.append(torch.cuda.Stream(device=JT.device, priority=-1 if len(self._group_streams) < ngroups - 1 else 0))
        side, T2 = self._side_stream, nat.GRAM_TILE * nat.GRAM_TILE
        with self._Section(self, "gram"):
            start = torch.cuda.Event()
            start.record(main)
            readies = []
            for g in range(ngroups):
                r0, r1 = bounds[g], bounds[g + 1]
                cs = self._group_streams[g] if concurrent else main
                if concurrent:
                    cs.wait_event(start)
                with torch.cuda.stream(cs):
                    nat.call("vmcpde_gram_rows", nat.ptr(JT), ldj, Pg, n, r0, r1, nat.ptr(b["packed"]), nat.ptr(b["gram_ws"]), b["gram_ws"].numel() * 8,
                             nat.stream())
                ready = torch.cuda.Event()
                ready.record(cs)
                readies.append(ready)
                side.wait_event(ready)
                piece = b["pack"][tri(r0) * T2:(tri(r1) * T2 if g + 1 < ngroups else b["pack"].numel())]
                with torch.cuda.stream(side):
                    mpi.allreduce_sum_(piece)
            if concurrent:
                for ready in readies:
                    main.wait_event(ready)
        with self._Section(self, "allreduce"):       # what is left of the collective when the last group's tiles are done: its exposed part
            done = torch.cuda.Event()
            done.record(side)
            main.wait_event(done)
        self.allreduceInfo = dict(overlapped=True, groups=ngroups, tile_rows=bounds, group_streams=concurrent)

    def _gram_launch(self, b, JT, ldj, f32, P, n, c, w, out):
        """The fp64-MFMA syrk (csrc/gram.hip) on the parameter-major Jacobian: packed lower-triangle tiles of sum_n w_n^2 (J_n - c)(J_n - c)^T."""
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
                    self._ratio_prev = float(self._ratio_host[0])      # (also what the low-rank routes' tolerance reads: fixed for this call)
                    self._uncentred_ok = bool(self._ratio_prev < 1e3)
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
            self._gram_and_allreduce(b, JT, ldj, P + 2 if bordered else P, n, c_gram)
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
            if self.gramShift == "auto":     # asynchronous safety check for the next call: rides in the call's one health record (_record_guard)
                if self._ratio_queued is not None:       # (a caller that never reached solve(): the figure of that call goes out on its own)
                    self._record_guard(None, self._guard_chol, keep_guard=True)
                self._ratio_queued = scal
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
        second moment behind rhoVar, tdvp.py:68-70; w = None: S0 itself)."""
        z = self._lazy
        if z["owner"] is not None and z["owner"].workspace_generation(z["n"]) != z["gen"]:
            raise RuntimeError("SExp / rhoVar / snr are computed lazily from the Jacobian of the last TDVP call, but the VarState "
                               "has evaluated another batch of the same size since (its workspace was overwritten): read these "
                               "attributes before calling psi(...), psi.hessian or psi.evaluate again")
        P, device = z["P"], z["JT"].device
        b = z["b"] if z["b"] is not None else self._buffers(P, device)
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
    def V(self):
        """tdvp.py:59-64: the eigenvectors of S (columns, ascending eigenvalues).  After the rank-revealing decomposition the P - r columns
        of the null space are formed on first access (lowrank.py: a call itself only needs the r Ritz vectors)."""
        if self._V is None and self._V_lazy is not None:
            with torch.cuda.device(self._ritz["U"].device):
                self._V = self._V_lazy()
            self._V_lazy = None
        return self._V

    @V.setter
    def V(self, value):
        self._V, self._V_lazy = value, None

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
            if self.ev is None:
                raise RuntimeError("rhoVar/snr need the eigenbasis: use solver='eigh'")
            rz = getattr(self, "_ritz", None)
            if rz is not None and self._lazy["JT"].dtype == global_defs.tReal and os.environ.get("VMCPDE_RHOVAR_LOWRANK", "1") != "0":
                self._rhoVar = self._rhoVar_from_ritz(rz)
            else:
                G = self._weighted_gram(self._lazy["e"] - self._lazy["Emean"])
                self._rhoVar = ((G @ self.V) * self.V).sum(0) - self.VtF ** 2
        return self._rhoVar

    def _rhoVar_from_ritz(self, rz):
        """tdvp.py:68-70 after the rank-revealing decomposition: (V^T G V)_kk = (1/N) sum_n E~_n^2 (O~_n . v_k)^2, and a vector with
        v^T S0 v = (1/N) sum_n (O~_n . v)^2 = 0 is orthogonal to EVERY sample's centred gradient -- so the P - r eigenvectors the engine
        reports with eigenvalue exactly 0 have rhoVar = 0 (and VtF = 0), and only the r Ritz vectors U = Q1 W need the samples:
        Z = U^T (J - m 1^T), one r x P x N product over the retained Jacobian (114 GFLOP at the headline size instead of the 1.1 TFLOP
        weighted Gram and the P^3 product behind it: 2.5 ms instead of 20 ms for the `snr` the reference's driver logs after every step,
        main.py:187).  Modes the engine reports as null carry rhoVar = 0, snr = 0."""
        z = self._lazy
        if z["owner"] is not None and z["owner"].workspace_generation(z["n"]) != z["gen"]:
            raise RuntimeError("SExp / rhoVar / snr are computed lazily from the Jacobian of the last TDVP call, but the VarState "
                               "has evaluated another batch of the same size since (its workspace was overwritten): read these "
                               "attributes before calling psi(...), psi.hessian or psi.evaluate again")
        P, n, r = z["P"], z["n"], rz["r"]
        U, wv = rz["U"], rz["wv"]
        pos = rz.get("pos")
        Jv = z["JT"][:P, :n]                                          # P x n view of the parameter-major Jacobian (row pitch ldj)
        Z = U.T @ Jv                                                  # r x n
        Z -= (U.T @ z["mean"])[:, None]
        Z *= (z["e"] - z["Emean"])[None, :]
        second = (Z * Z).sum(1)                                       # sum_n E~_n^2 (O~_n . u_k)^2 over this rank's samples
        mpi.allreduce_sum_(second)
        if pos is None:
            pos = torch.arange(r, device=U.device) + (P - r) * (wv >= 0)   # the Ritz columns of V: negatives first, the positives behind the zeros
        out = torch.zeros(P, dtype=global_defs.tReal, device=U.device)
        out[pos] = second / z["Ntot"] - self.VtF[pos] ** 2
        return out

    @property
    def snr(self):
        """tdvp.py:71."""
        if self._snr is None:
            if self.ev is None:
                raise RuntimeError("rhoVar/snr need the eigenbasis: use solver='eigh'")
            rho = self.rhoVar
            snr = torch.sqrt(torch.abs(self._lazy["Ntot"] * (self.VtF * self.VtF) / rho))
            # (modes the rank-revealing engine reports as null have rhoVar = 0 exactly: no sample's gradient reaches them -- snr 0, not 0 / 0)
            self._snr = torch.where(rho == 0, torch.zeros_like(snr), snr) if getattr(self, "_ritz", None) is not None else snr
        return self._snr

    def _solver(self):
        return "eigh" if self.solver == "auto" else self.solver

    # ---- Cholesky health: pinned pivots and the solver residual ------------------------------------------------------
    # slots of the health record (8-byte words): [0:4] residual, TDVP error, pinned pivots, spare (vmcpde_tdvp_diagnostics) | [4] max_p mean_p^2 / var_p
    # of this call's S-build | [5] (rank, done) of the low-rank factorisation (two int32) | [6:10] head of its workspace (d0, trace left, step counts)
    # | [10:13] a factorisation on the frozen pivot set: trace(R) / min D, pinned pivots, max R_ii
    _REC_WORDS = 16

    def _record_guard(self, out, chol, keep_guard=False):
        """Everything the host wants to know about this call without waiting for it -- (residual, pinned-pivot count) of the solve, the mean^2 / var
        figure of the S-build, rank / done / step count of the low-rank factorisation on a static plan -- gathered into ONE device buffer by
        `vmcpde_gather_words` and queued as ONE asynchronous copy into a pinned record; looked at by `_check_guard` at the next host-side guard (the
        NaN check of `__call__`, `check()` -- called by the steppers before a step is accepted --, or the start of the next right-hand side) and by
        the choices the next call makes (centred / uncentred operands, re-planning of the static route)."""
        dev = (out if out is not None else self._ratio_queued).device
        rec = getattr(self, "_rec", None)
        if rec is None or rec["dev"].device != dev:
            host = torch.zeros(self._REC_WORDS, dtype=global_defs.tReal).pin_memory()
            rec = self._rec = dict(dev=torch.zeros(self._REC_WORDS, dtype=global_defs.tReal, device=dev), host=host, event=torch.cuda.Event(),
                                   srcs=(nat.C.c_void_p * 8)(), counts=(nat.C.c_int32 * 8)())
            self._guard_host, self._guard_event = host[0:4], rec["event"]
        static = self._static_queued
        parts = [(out, 4), (None if self._ratio_queued is None else self._ratio_queued[4:5], 1),
                 (None if static is None else static[0]["state"], 1), (None if static is None else static[0]["ws"], 4),
                 (static[0]["finfo"] if static is not None and static[2] else None, 3)]
        for i, (src, n) in enumerate(parts):
            rec["srcs"][i] = None if src is None else src.data_ptr()
            rec["counts"][i] = n
        with torch.cuda.device(dev):
            nat.call("vmcpde_gather_words", rec["srcs"], rec["counts"], len(parts), nat.ptr(rec["dev"]), nat.stream())
            rec["host"].copy_(rec["dev"], non_blocking=True)
            rec["event"].record()
        if self._ratio_queued is not None:
            self._ratio_host, self._ratio_event = rec["host"][4:5], rec["event"]
            self._ratio_queued = None
        if static is not None:           # the static plan reads (rank, done) and the step count of THIS factorisation at the next call
            st = static[1]
            st["host"], st["head"], st["event"] = rec["host"][5:6].view(torch.int32), rec["host"][6:10], rec["event"]
            st["finfo"], st["frozen_last"] = rec["host"][10:13], bool(static[2])
            self._static_queued = None
        if not keep_guard:
            self._guard_chol = chol
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
            # the spectral core's stedc status (lowrank.py::_core_eigh_tridiag): its copy was queued before this record's, so it has landed.
            # A non-zero word means (ev, U) -- hence VtF and the update -- came from an unconverged decomposition: finite but wrong, so the
            # step is rejected like torch.linalg.eigh rejects a non-zero info, and the core falls back to the library solver
            st = getattr(self, "_tridiag", None)
            if st is not None and st.get("pending"):
                st["pending"], st["event"] = False, None
                code = int(st["host"][0])
                if code != 0:
                    st["off"] = True
                    raise FloatingPointError(f"the spectral solve's tridiagonal eigensolver (stedc) reported {code} unconverged block(s) ({when}): "
                                             f"the update of that call is not trustworthy; later calls use the library eigensolver")
            if math.isnan(resid):
                raise FloatingPointError(f"nan encountered in the TDVP update ({when})")
            return
        if not (resid <= self.residualBound):      # also catches NaN
            if (self.choleskyInfo or {}).get("mode") == "static":
                # the rank of S outgrew the plan of the low-rank solve within one call (it is re-planned 32 columns before the pad, and a
                # factorisation that stops at the pad still refines to ~1e-8: this needs a jump of the rank by a large factor): the
                # plan is dropped, so a caller that catches this and repeats the step gets a planning call
                self._drop_plans()
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

    def _drop_plans(self):
        """A static plan failed its residual guard: the next call plans anew."""
        self._chol_static = None

    # ---- tdvp.py:57-71 -------------------------------------------------------------------------------------------
    def transform_to_eigenbasis(self, S, F, EOdata=None):
        P = F.numel()
        done = False
        on_device = (P >= DEVICE_EIGH_MIN_P) if self.diagonalizeOnDevice is None else bool(self.diagonalizeOnDevice)
        if self.spectralEngine == "lowrank" and self.diagonalizeOnDevice is None:
            on_device = True
        self.spectralInfo = None
        self._ritz = None                # (set by the rank-revealing engine: what rhoVar / snr need of it)
        self._V = self._V_lazy = None
        # (rejected by the svdTol gate before: a LARGER svdTol passes at once; with the same one the attempt is repeated only after
        # LOWRANK_GATE_RETRY_AFTER calls -- the gate compares the factorisation's remainder with svdTol x lambda_max of the CURRENT S, whose
        # spectrum moves along a trajectory, so the rejection is long-lived but not permanent)
        gated = self.spectralEngine == "auto" and getattr(self, "_lowrank_gate_svdtol", None) is not None and \
            float(self.svdTol) <= self._lowrank_gate_svdtol
        if gated and on_device:
            self._lowrank_gate_retry = getattr(self, "_lowrank_gate_retry", LOWRANK_GATE_RETRY_AFTER) - 1
            if self._lowrank_gate_retry <= 0:
                gated, self._lowrank_gate_svdtol = False, None
        lowrank = on_device and self.spectralEngine != "dense" and not (self.diagonalShift > 1e-10) and self._lowrank_tol() is not None and \
            (self.spectralEngine == "lowrank" or (P >= LOWRANK_MIN_P and self._lowrank_backoff == 0 and not gated))
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
                lazy = self._V_lazy                       # (set by the engine; the setter of V clears it)
                self.ev, self.V = res
                self._V_lazy = lazy
                done = True
            elif self.spectralEngine == "auto":
                if (self.spectralInfo or {}).get("rejected"):
                    # the svdTol gate: a property of (svdTol, this model's spectrum) rather than of the call -- the wasted factorisation would
                    # recur at every retry, so the attempt stays off until svdTol is raised or LOWRANK_GATE_RETRY_AFTER calls have passed
                    self._lowrank_gate_svdtol = float(self.svdTol)
                    self._lowrank_gate_retry = LOWRANK_GATE_RETRY_AFTER
                else:
                    self._lowrank_backoff = LOWRANK_RETRY_AFTER    # rank above the cap: dense for the next calls
        if on_device and not done:
            if self.spectralInfo is None or self.spectralInfo.get("engine") != "lowrank":
                self.spectralInfo = dict(engine="dense")
            else:
                self.spectralInfo = dict(self.spectralInfo, engine="dense", tried="lowrank")
            if P >= DENSE_EIGH_WARN_P and not getattr(self, "_warned_dense_eigh", False):
                import warnings
                why = ("the rank-revealing engine was tried and does not apply to this S: full numerical rank, or rejected by the svdTol gate"
                       if (self.spectralInfo or {}).get("tried") else "diagonalShift > 0, spectralEngine='dense', or the engine is backing off")
                warnings.warn(f"spectral solve of a {P} x {P} S by the dense eigensolver (rocSOLVER syevd; {why}): about 150 ms per call at 4155 "
                              f"parameters and 2.6 s at 16 181, replicated on every rank -- 13x the S-build's share of a sharded step "
                              f"(profiles/r05_run2_bench_C4_eigh.json).  solver='cholesky' solves the regularised system in 2.3 / 36 ms instead; it has "
                              f"no eigenbasis (no ev / snr).  This warning is issued once per TDVP object.", RuntimeWarning, stacklevel=3)
                self._warned_dense_eigh = True
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
            self.spectralInfo = dict(self.spectralInfo or {}, engine="host")       # (the reference's default: LAPACK on the host)
        if self._V is None and self._V_lazy is not None:
            # the rank-revealing engine: V^T F from the r Ritz vectors, exact zeros on the null modes (lowrank.py)
            rz = self._ritz
            self.VtF = torch.zeros(P, dtype=global_defs.tReal, device=F.device)
            vr = torch.empty(rz["r"], dtype=global_defs.tReal, device=F.device)
            with torch.cuda.device(F.device):
                nat.call("vmcpde_gemv", nat.ptr(rz["U"]), rz["U"].stride(0), P, rz["r"], 1, nat.ptr(F.contiguous()), nat.ptr(vr), nat.stream())   # tdvp.py:66
            self.VtF[rz["pos"]] = vr
            return
        self.VtF = torch.empty(P, dtype=global_defs.tReal, device=F.device)
        Vc = self.V.contiguous()
        self.V = Vc
        with torch.cuda.device(F.device):
            nat.call("vmcpde_gemv", nat.ptr(Vc), P, P, P, 1, nat.ptr(F.contiguous()), nat.ptr(self.VtF), nat.stream())   # tdvp.py:66

    # ---- tdvp.py:73-94 -------------------------------------------------------------------------------------------
    def solve(self, Eloc, gradients, logProbs, psi=None):
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
                if self._V is None and self._V_lazy is not None:       # (null modes carry the coefficient 0: only the Ritz vectors count)
                    rz = self._ritz
                    cr = coef[rz["pos"]].contiguous()
                    nat.call("vmcpde_gemv", nat.ptr(rz["U"]), rz["U"].stride(0), P, rz["r"], 0, nat.ptr(cr), nat.ptr(update), nat.stream())   # tdvp.py:91
                else:
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
        """Everything a call inherits from the previous one: the global means used as the shift of the one-pass moments and the Gram
        (centering / gramShift "auto") with the cancellation figure that goes with them, and -- on the low-rank routes, which the
        Cholesky solve takes from LOWRANK_CHOL_MIN_P and the spectral solve from LOWRANK_MIN_P parameters on -- the static plan and the
        pivot list of the last factorisation (lowrank.py).  With it a resumed run repeats the uninterrupted one bit for bit
        (examples/main.py; tested at P = 37 on the dense route and at P = 4213 on the static low-rank route)."""
        if self._ratio_event is not None:
            self._ratio_event.synchronize()
            self._ratio_prev = float(self._ratio_host[0])
            self._uncentred_ok = bool(self._ratio_prev < 1e3)
        return {"c_prev": None if self._c_prev is None else self._c_prev.detach().cpu().clone(),
                "cE_prev": None if self._c_prev is None else self._cE_prev.detach().cpu().clone(),
                "uncentred_ok": bool(self._uncentred_ok), "ratio_prev": self._ratio_prev,
                "lowrank": self._lowrank_state_dict()}

    def load_state_dict(self, st, device=None):
        device = global_defs.my_device() if device is None else device
        self._c_prev = None if st.get("c_prev") is None else st["c_prev"].to(device)
        self._cE_prev = None if st.get("cE_prev") is None else st["cE_prev"].to(device)
        self._uncentred_ok = bool(st.get("uncentred_ok", False))
        self._ratio_prev = st.get("ratio_prev")
        self._ratio_host = self._ratio_event = None      # the restored figures stand until the next call's own check lands
        if st.get("lowrank") is not None:
            with torch.cuda.device(device):
                self._lowrank_load_state(st["lowrank"])

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

    @staticmethod
    def _fused_sample_eval(psi, evolutionEq):
        """Sampling + evaluation in one flow launch (`evolutionEq.from_latent`)?  Only for the package's own objects used as they come: a `psi` whose
        `sample` / `evaluate` a caller has replaced (tests record what was drawn; a driver may hand its own samples) or an `evolutionEq` of another
        type keeps the reference's two calls.  VMCPDE_FUSED_SAMPLE_EVAL=0: the two launches (A/B)."""
        if os.environ.get("VMCPDE_FUSED_SAMPLE_EVAL", "1") == "0":
            return False
        from .var_state import VarState
        from .evolutionEq import EvolutionEquation
        if type(evolutionEq) is not EvolutionEquation or not isinstance(psi, VarState) or getattr(psi, "keep_activations", False):
            return False
        if getattr(psi.net, "global_change", False):      # (net.py:150-151: the reference's inverse direction is not the inverse map there)
            return False
        overridden = {"sample", "evaluate", "flow_sample", "draw_latent", "evaluate_from_latent", "__call__"} & set(vars(psi))
        same_class = all(getattr(type(psi), name) is getattr(VarState, name) for name in ("sample", "evaluate", "flow_sample", "draw_latent", "evaluate_from_latent"))
        return not overridden and same_class

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
            if self._fused_sample_eval(psi, evolutionEq):
                # tdvp.py:117 + :122 with ONE flow launch: the sampler's latent points go straight to the evaluation, whose reverse sweep produces
                # the samples x on its way from the latent end (VarState.evaluate_from_latent) -- no separate inverse-flow launch
                start_timing("sampling")
                latent = psi.draw_latent(nSamplesTDVP)
                stop_timing("sampling")
                start_timing("compute Eloc")
                sampleConfigs, Eloc, sampleGradients, logProbs = evolutionEq.from_latent(psi, latent, t)
                stop_timing("compute Eloc")
            else:
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