"""The pivoted low-rank factor of S and the two solve routes built on it (reference: vmc_fluids/tdvp.py:57-94).

S = <J^T J> of these models is numerically rank deficient (238 of 4213 eigenvalues above 1e-16 lambda_max at the headline size), so
both of the reference's systems can be solved from S = L L^T + R with a P x r factor (csrc/eig.hip: blocked pivoted Cholesky):
  * `_solve_lowrank` / `_solve_lowrank_static`: solver="cholesky" -- (S + D) u = F through the Woodbury identity and iterative
    refinement against the full matrix;
  * `_eigh_lowrank`: solver="eigh" -- (ev, V) of tdvp.py:57-64 from a Householder basis of range(L) and a Rayleigh-Ritz step.
`LowRankRoutes` is mixed into `tdvp.TDVP`, which decides when a route applies and falls back to the dense factorisations.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _native as nat
from . import global_defs


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
LOWRANK_SVDTOL_FRACTION = 0.1  # spectral route: ... and below this x svdTol x (a lower bound of lambda_max).  What the factorisation drops is
                            #   reported as exact zeros; the reference keeps every mode above its hard cut at 1e-14 lambda_max and weights it by
                            #   1 / (1 + (svdTol lambda_max / lambda)^6) (tdvp.py:82-85): a dropped mode below 0.1 svdTol lambda_max had a weight
                            #   below 1e-6, a smaller svdTol sends the call to the dense decomposition
LOWRANK_CHOL_MIN_P = 4096   # choleskyEngine="auto": the low-rank solve from this many parameters on.  P = 4213: 1.3 ms against 2.1 ms for the blocked
                            #   factorisation of the full matrix (which scales as P^3: 0.8 ms at 3072) -- in the static mode, i.e. without a host
                            #   read in the step; with one (planning mode) 2.5 ms.  P = 16 181: 7 against 38 ms
LOWRANK_SOLVE_RHO = 1e-4    # low-rank Cholesky solve: largest trace(R) / min D accepted (the contraction factor of its iterative refinement)
LOWRANK_RETRY_AFTER = 50    # "auto": calls on the dense path after one whose rank exceeded the cap, before the next attempt
FROZEN_REFRESH_AFTER = 7    # static plan: consecutive calls that factor S on the PREVIOUS pivot set (no search) before a pivoted factorisation refreshes
                            #   the set -- earlier when the remainder it measures (trace(R) / min D, reported one call later) outgrows LOWRANK_SOLVE_RHO
LOWRANK_GATE_RETRY_AFTER = 8 * LOWRANK_RETRY_AFTER   # ... after one the svdTol gate rejected (the spectrum moves along a trajectory, slowly)


def _round_up(v, m):
    return (v + m - 1) // m * m


def frozen_set_decision(enabled, prev_rank, frozen_run, last_was_frozen, last_record):
    """Does THIS call on a static plan factor S on the previous pivot set (True) or run the pivoted factorisation (False)?  A pure function of host
    values every rank holds identically: the switch, the rank the last factorisation reported, how many calls in a row have already used the set
    (at most FROZEN_REFRESH_AFTER), and -- when the last call was such a call -- the remainder it measured, `last_record` = (trace(R) / min D, pinned
    pivots): above the refinement's contraction bound LOWRANK_SOLVE_RHO, not finite, or with a pinned pivot, the set no longer spans S."""
    if not enabled or prev_rank < 1 or frozen_run >= FROZEN_REFRESH_AFTER:
        return False
    if last_was_frozen:
        rho, pinned = last_record
        return bool(pinned == 0.0 and rho <= LOWRANK_SOLVE_RHO)          # (a NaN fails the comparison)
    return True


class StaticPlanOutdated(FloatingPointError):
    """The numerical rank of S outgrew the static plan of the low-rank solve within one call, so that call's update failed the residual
    guard.  The plan has been dropped: repeating the step (the steppers do, once: `retry_step`) runs a planning call."""
    retry_step = True



class LowRankRoutes:
    """Mix-in of `tdvp.TDVP`: state `_chol_static` (the static plan of the Cholesky route), `_lowrank_last_rank`, `_buf_lr` (factor,
    pivot list and workspace per (P, device)); reads `svdTol`, `diagonalShift`, `pivotHints`, `choleskyEngine`, `spectralEngine`,
    `_floor`, `_uncentred_ok`, `_ratio_prev`; writes `choleskyInfo` / `spectralInfo`."""

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
            saved = getattr(self, "_lr_restore", None)           # a resumed run: the pivot list and step count of the interrupted one
            if saved is not None and saved["P"] == P and saved["piv"].numel() <= cap:
                w["piv"][:saved["piv"].numel()] = saved["piv"].to(device)
                w["state"].copy_(saved["state"].to(device))
                w["head_host"].copy_(saved["head"])
                # (the head of the device workspace too -- d0, trace left, step counts: a call on the frozen pivot set does not rewrite it, and the
                # record it sends to the host must carry the last pivoted factorisation's step count as in the uninterrupted run)
                w["ws"][:4].copy_(saved.get("ws_head", saved["head"]).to(device))
            self._lr_restore = None
        return w

    def _lowrank_tol(self):
        """Pivoting tolerance of the low-rank routes for THIS call's S, or None when they do not apply: the factorisation must stop
        above the noise of S.  An fp64-accumulated Gram (fp64 or fp32 Jacobian storage) is exact to round-off: LOWRANK_TOL.  fp32 block
        sums (2e-8) bury everything below ~1e-8 max diag in noise -- such an S has no usable numerical rank, so that variant keeps the
        dense paths.  A function of values fixed before the call's first launch (`_ratio_prev` is the PREVIOUS call's mean^2 / var
        figure, read after its event at the top of get_tdvp_equation): every launch of a factorisation, and every rank, sees the same."""
        if self.f32BlockGram:
            return None
        tol = LOWRANK_TOL
        r = getattr(self, "_ratio_prev", None)
        if self._uncentred_ok and r is not None and r == r:
            # the uncentred Gram subtracts N m m^T afterwards: relative noise ~ eps * max_p mean_p^2 / var_p (the figure the engine
            # watches to stay below 1e3; 1.3e-4 for the headline network)
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
        max diag, trace of R, extra's values).  The first call covers the previous call's rank (the spectrum moves slowly along a
        trajectory); the host reads `state` after every call (`extra`: a function returning up to two device scalars, evaluated behind
        the first launch chain and copied along with that read)."""
        P, device = S.shape[0], S.device
        done, first = False, True
        tol = self._lowrank_tol() or LOWRANK_TOL                   # one value for every launch of this factorisation
        self._arm_hints(w, P)
        # (first launch chain: what the previous factorisation in these buffers used plus a third; an unhinted step accepts ~13 columns)
        used = self._steps_used(w["head_host"])
        nblocks = used + max(3, used // 3) if used else ((self._lowrank_last_rank + 12) // 13 + 2 if self._lowrank_last_rank else 12)
        if w["info_host"].numel() < 4:
            w["info_host"] = torch.zeros(4, dtype=global_defs.tReal).pin_memory()
        n_extra = 0
        for _ in range(cap // 3 + 8):           # (every step makes progress -- a column, or candidates lowered to what is left of them)
            nat.call("vmcpde_pivoted_cholesky_blocked", nat.ptr(S), S.stride(0), P, tol, 1 if first else 0, nblocks, cap, nat.ptr(w["LT"]),
                     w["ldl"], nat.ptr(w["piv"]), nat.ptr(w["state"]), nat.ptr(w["ws"]), w["ws"].numel() * 8, nat.stream())
            w["state_host"].copy_(w["state"], non_blocking=True)
            w["info_host"][:2].copy_(w["ws"][:2], non_blocking=True)
            w["head_host"].copy_(w["ws"][:4], non_blocking=True)
            if extra is not None and first:
                ex = extra().reshape(-1)
                n_extra = ex.numel()
                w["info_host"][2:2 + n_extra].copy_(ex, non_blocking=True)
            torch.cuda.current_stream(device).synchronize()
            first, nblocks = False, 3
            if int(w["state_host"][1]):
                done = True
                break
            if int(w["state_host"][0]) >= cap:
                break
        r = int(w["state_host"][0])
        done = done or r == P                                       # full rank: every row was a pivot, nothing is left
        return (r, done, float(w["info_host"][0]), float(w["info_host"][1]), *[float(w["info_host"][2 + i]) for i in range(n_extra)])

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
        st = self._chol_static
        if st is not None and st["P"] == P and st["rmax"] <= w["LT"].shape[0]:
            prev_rank = st["rank"]
            if st["event"] is not None:                 # the previous static call's factorisation: finished, with room to spare?
                st["event"].synchronize()
                prev_rank, prev_done = int(st["host"][0]), int(st["host"][1])
                if not prev_done or prev_rank > st["rmax"] - 32:     # (re-plan well before the rank reaches the pad)
                    st = self._chol_static = None
                else:                                                # what the last factorisation used, and a third (at least 3) to spare
                    used = self._steps_used(st["head"])
                    st["nsteps"] = used + max(3, used // 3)
            if st is not None:
                # Factor on the previous pivot set, without the search (vmcpde_frozen_gather / vmcpde_cholesky_tall / vmcpde_frozen_finish)?  Yes while
                # the set is fresh -- at most FROZEN_REFRESH_AFTER calls since the last pivoted factorisation -- and the last such call's remainder,
                # read here one call later, was small (trace(R) / min D within the refinement's contraction bound, no pinned pivot).  A function of
                # host values every rank holds identically.
                enabled = bool(getattr(self, "frozenPivots", True)) and os.environ.get("VMCPDE_FROZEN_PIVOTS", "1") != "0"
                frozen = frozen_set_decision(enabled, prev_rank, st.get("frozen_run", 0), bool(st.get("frozen_last")),
                                             None if not st.get("frozen_last") else (float(st["finfo"][0]), float(st["finfo"][1])))
                st["frozen_run"] = st.get("frozen_run", 0) + 1 if frozen else 0
                self._solve_lowrank_static(b, F, P, update, w, st, frozen_rank=prev_rank if frozen else None)
                return True
        D = b["diagS"] * float(self.diagonalShift)                 # (planning call: array expressions; a static call forms D in vmcpde_woodbury_prepare)
        if self._floor is not None:
            D = D + self._floor
        r, done, d0, left, dmin = self._pivoted_factor(S0, w, cap, extra=lambda: D.min())
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

    def _solve_lowrank_static(self, b, F, P, update, w, st, frozen_rank=None):
        """A call on the static plan: own launches only -- the hinted factorisation, `vmcpde_woodbury_prepare` (D, 1 / D, the zero rows behind the
        rank), `vmcpde_woodbury_gram` (G = I + L^T D^-1 L on the matrix pipe), the blocked Cholesky of G, and per application of M^-1 a projection,
        the r x r substitution in place and a combination -- no library GEMM, no framework elementwise kernel, and no copy: rank / done / step count
        of the factorisation reach the host in the call's one health record (`TDVP._record_guard`)."""
        device, rmax = F.device, st["rmax"]
        S0 = b["S"][:P, :P]
        ldl = w["ldl"]
        if w.get("G") is None or w["G"].shape[0] != rmax:
            need = nat.lib().vmcpde_chol_workspace_bytes(rmax)
            w["G"] = torch.zeros((rmax, rmax), dtype=global_defs.tReal, device=device)
            w["G_ws"] = torch.empty(need // 8, dtype=global_defs.tReal, device=device)
            w["G_part"] = torch.empty(nat.lib().vmcpde_woodbury_gram_workspace_bytes(ldl, rmax) // 8, dtype=global_defs.tReal, device=device)
            w["Su"] = torch.empty(P, dtype=global_defs.tReal, device=device)
        if w.get("D") is None or w["D"].numel() != P:
            w["D"] = torch.empty(P, dtype=global_defs.tReal, device=device)
            w["invD"] = torch.empty(ldl, dtype=global_defs.tReal, device=device)
        G, gws, Su, D = w["G"], w["G_ws"], w["Su"], w["D"]
        need = gws.numel() * 8
        if frozen_rank is not None:
            # S = L L^T + R on the previous factorisation's pivot set (w["piv"][:r], in its order): gather [S[Pi, Pi]; S[:, Pi]], one tall Cholesky (two
            # panels at the headline size instead of ~10 block steps of two launches with their arg-max searches), transpose into the factor's layout
            r = int(frozen_rank)
            if w.get("T") is None or w["T"].shape[1] != rmax:
                w["T"] = torch.zeros((_round_up(rmax + P, 128), rmax), dtype=global_defs.tReal, device=device)     # (pad rows: read by the strips, never written)
                w["T_ws"] = torch.empty(nat.lib().vmcpde_chol_workspace_bytes(rmax) // 8, dtype=global_defs.tReal, device=device)
                w["left"] = torch.empty(P, dtype=global_defs.tReal, device=device)
                w["finfo"] = torch.zeros(3, dtype=global_defs.tReal, device=device)
                w["T_info"] = torch.zeros(1, dtype=torch.int32, device=device)
            T = w["T"]
            nat.call("vmcpde_frozen_gather", nat.ptr(S0), S0.stride(0), P, nat.ptr(w["piv"]), r, rmax, nat.ptr(T), rmax, nat.stream())
            nat.call("vmcpde_cholesky_tall", nat.ptr(T), rmax, rmax + P, rmax, nat.ptr(w["T_info"]), nat.ptr(w["T_ws"]), w["T_ws"].numel() * 8, nat.stream())
            nat.call("vmcpde_frozen_finish", nat.ptr(T), rmax, nat.ptr(S0), S0.stride(0), P, rmax, nat.ptr(w["LT"]), ldl, nat.ptr(w["left"]), nat.stream())
        else:
            self._arm_hints(w, P)                                   # (rmax is the padded size of G; a small P caps the rank below it)
            nat.call("vmcpde_pivoted_cholesky_blocked", nat.ptr(S0), S0.stride(0), P, self._lowrank_tol() or LOWRANK_TOL, 1, st["nsteps"], min(rmax, P), nat.ptr(w["LT"]), ldl,
                     nat.ptr(w["piv"]), nat.ptr(w["state"]), nat.ptr(w["ws"]), w["ws"].numel() * 8, nat.stream())
        # (rank, done), the step count and -- frozen set -- the measured remainder: for the next call, in this call's health record
        self._static_queued = (w, st, frozen_rank is not None)
        # D = diagonalShift diag(S0) + floor, 1 / D; rows of LT past the rank: zero, i.e. an identity block in G (the factorisation rewrites rows
        # below the rank in full -- pad columns included --, the rows behind it hold an earlier call's columns)
        nat.call("vmcpde_woodbury_prepare", nat.ptr(b["diagS"]), float(self.diagonalShift), nat.ptr(self._floor) if self._floor is not None else None, P, ldl,
                 nat.ptr(D), nat.ptr(w["invD"]), nat.ptr(w["LT"]), nat.ptr(w["state"]), rmax, nat.stream())
        if frozen_rank is not None:                                 # trace(R) / min D, pinned pivots, max R_ii; state = (r, done) as a pivoted run leaves it
            nat.call("vmcpde_frozen_check", nat.ptr(w["left"]), P, nat.ptr(D), int(frozen_rank), nat.ptr(w["T_info"]), nat.ptr(w["state"]), nat.ptr(w["ws"]),
                     nat.ptr(w["finfo"]), nat.stream())
        # G = I + L^T D^-1 L, contraction length P, result only rmax x rmax: 64 x 64 blocks x chunks of 256 parameters as independent workgroups,
        # the scaling folded into the operand load (as ONE library GEMM the shape runs on 8 workgroups, 0.23 ms; round 5's batched GEMM + sum +
        # division took three launches)
        nat.call("vmcpde_woodbury_gram", nat.ptr(w["LT"]), ldl, rmax, nat.ptr(w["invD"]), nat.ptr(G), rmax, nat.ptr(w["G_part"]), w["G_part"].numel() * 8,
                 nat.stream())
        # (its status word -- pinned pivots, i.e. a G that is not positive definite: a non-finite factor -- is the call's `info`, which
        # vmcpde_tdvp_diagnostics hands to the residual guard like the dense factorisation's)
        nat.call("vmcpde_cholesky", nat.ptr(G), rmax, rmax, nat.ptr(b["info"]), nat.ptr(gws), need, nat.stream())

        # u = M^-1 F, then u <- u + M^-1 (F - (S0 + D) u) against the TRUE matrix: per application two own launches around the r x r
        # substitution (vmcpde_woodbury_project / _combine; the residual is formed inside the first; the substitution runs in place in t1)
        if w.get("wb_y") is None or w["wb_y"].numel() != P or w["wb_t1"].numel() != rmax:
            w["wb_y"] = torch.empty(P, dtype=global_defs.tReal, device=device)
            w["wb_t1"] = torch.empty(rmax, dtype=global_defs.tReal, device=device)
        yb, t1 = w["wb_y"], w["wb_t1"]
        Fc = F if F.is_contiguous() else F.contiguous()

        def apply(u_in):
            nat.call("vmcpde_woodbury_project", nat.ptr(w["LT"]), ldl, P, rmax, nat.ptr(D), nat.ptr(Fc), nat.ptr(Su) if u_in is not None else None,
                     nat.ptr(u_in), nat.ptr(yb), nat.ptr(t1), nat.stream())
            nat.call("vmcpde_chol_solve", nat.ptr(G), rmax, rmax, nat.ptr(t1), nat.ptr(t1), nat.ptr(gws), need, nat.stream())
            nat.call("vmcpde_woodbury_combine", nat.ptr(w["LT"]), ldl, P, rmax, nat.ptr(D), nat.ptr(yb), nat.ptr(t1), nat.ptr(u_in), nat.ptr(update),
                     nat.stream())

        apply(None)
        for _ in range(st["nref"]):
            nat.call("vmcpde_gemv", nat.ptr(b["S"]), b["Ppad"], P, P, 0, nat.ptr(update), nat.ptr(Su), nat.stream())
            apply(update)
        self.choleskyInfo = dict(engine="lowrank", mode="static", rmax=rmax, block_steps=0 if frozen_rank is not None else st["nsteps"], refinements=st["nref"],
                                 planned_rank=st["rank"], factor="frozen pivots" if frozen_rank is not None else "pivoted")

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
        # lambda_max from below, for the svdTol gate: the trace over the rank, and |L e_1|^2 = |S[:, p_1]|^2 / S[p_1, p_1] (the first column of
        # the factor is complete after the first block step) -- both <= lambda_max(L L^T) <= lambda_max(S)
        r, done, d0, left, tr, col1 = self._pivoted_factor(S, w, cap, extra=lambda: torch.stack([torch.diagonal(S).sum(), (w["LT"][0, :P] ** 2).sum()]))
        lam_lb = max(d0, tr / max(r, 1), col1) if np.isfinite(tr) and np.isfinite(col1) else d0
        self.spectralInfo = dict(engine="lowrank", rank=r, cap=cap, max_diag=d0, trace_left=left, converged=done, lambda_max_lower_bound=lam_lb)
        if not done or r < 1 or not (left <= LOWRANK_TRACE_BOUND * d0) or not np.isfinite(tr):
            return None
        if not (left <= LOWRANK_SVDTOL_FRACTION * float(self.svdTol) * lam_lb):
            # the reference would still give the dropped modes a visible weight with this svdTol (tdvp.py:82-85): dense decomposition
            self.spectralInfo["rejected"] = "trace_left above %g x svdTol x lambda_max" % LOWRANK_SVDTOL_FRACTION
            return None
        self._lowrank_last_rank = r
        core = self._core_choice(r)
        jacobi = core == "jacobi"
        Lc = w["LT"][:r].clone() if jacobi else None               # (the Householder chain overwrites the factor)
        Y, Xn, Q1 = self._lowrank_range_basis(w, P, r, device)
        SQ = S @ Q1
        Q1t = Q1.T.contiguous()
        T = _long_k_product(Q1t, SQ.T.contiguous())
        T = 0.5 * (T + T.T)
        if jacobi:
            wv, W = self._core_eigh_jacobi(w, T, _long_k_product(Lc[:, :P], Q1t), r, device)
        elif core == "tridiag":
            wv, W = self._core_eigh_tridiag(w, T, r, device)
        else:
            # (the r x r core on one host LAPACK thread takes 1.6 ms against 5.4 ms for syevd's launch chain at r = 238, but the host
            # synchronisation it needs costs the calling loop more than that: 33.0 against 30.2 ms per TDVP call -- measured and dropped)
            wv, W = torch.linalg.eigh(T)                           # r x r (rocSOLVER)
        self.spectralInfo["core"] = {"jacobi": "jacobi", "tridiag": "sytrd(own)+stedc+back-transform(own)"}.get(core, "syevd")
        U = Q1 @ W
        # Ascending order: the (few, round-off) negative Ritz values, the P - r zeros of the complement, the positive Ritz values -- as
        # device-side positions, so that nothing here waits for the host.  Everything a call needs of the eigenbasis lives in the r Ritz
        # vectors: V^T F vanishes on the complement (F is a combination of centred gradients, every one of them orthogonal to a vector with
        # v^T S v = 0), the pseudo-inverse gives those modes the coefficient 0 (tdvp.py:82), and rhoVar / snr vanish there (TDVP.rhoVar).
        # The P x (P - r) complement block of V (one more P x P x r product, 0.25 ms at the headline size) is therefore formed only when
        # somebody reads `tdvp.V` -- the reference's driver never does (main.py:157-209 stores ev and snr).
        pos = torch.arange(r, device=device) + (P - r) * (wv >= 0)
        ev = torch.zeros(P, dtype=global_defs.tReal, device=device)
        ev[pos] = wv
        self._ritz = dict(U=U, wv=wv, r=r, P=P, pos=pos)
        self._V_lazy = lambda: self._lowrank_assemble(Y, Xn, U, wv, r, P, device)[1]
        return ev, None

    # ---- the r x r core: which eigensolver ---------------------------------------------------------------------------------
    def _core_choice(self, r):
        """"tridiag" -- own Householder tridiagonalisation in one workgroup (csrc/tridiag.hip), rocSOLVER's stedc on the tridiagonal matrix,
        own back-transformation: what coreEigensolver="auto" (the default) takes while r fits the kernel (<= 256) and an earlier call's stedc
        has not reported a failure; "jacobi" -- the opt-in one-sided Jacobi kernel; "library" -- torch.linalg.eigh (rocSOLVER's syevd: 4.3 ms at
        r = 206, 2.9 ms of it its tridiagonalisation's launch chain).  VMCPDE_CORE_EIGH overrides the constructor argument."""
        want = os.environ.get("VMCPDE_CORE_EIGH") or getattr(self, "coreEigensolver", "auto")
        if want == "jacobi":
            return "jacobi" if self._core_uses_jacobi(r) else "library"
        if want in ("auto", "tridiag"):
            from . import rocsolver
            st = getattr(self, "_tridiag", None)
            if st is not None and st["event"] is not None and st.get("pending"):   # (a status nobody's guard has looked at yet, e.g. a direct
                st["event"].synchronize()                                          # use of the engine without TDVP.solve: read it here)
                st["event"], st["pending"] = None, False
                if int(st["host"][0]) != 0:
                    import warnings
                    warnings.warn(f"stedc reported {int(st['host'][0])} unconverged block(s) on the core of the spectral solve; using the library "
                                  f"eigensolver from now on", RuntimeWarning)
                    st["off"] = True
            ok = 3 <= r <= int(nat.lib().vmcpde_sytrd_medium_max_n()) and not (st is not None and st["off"]) and rocsolver.available()
            return "tridiag" if ok else "library"
        return "library"

    def _core_eigh_tridiag(self, w, T, r, device):
        """(wv ascending, W) with T W = W diag(wv): T = Q tridiag(d, e) Q^T by `vmcpde_sytrd_small` (one workgroup, the lower triangle in
        registers), the tridiagonal eigenproblem by rocSOLVER's divide & conquer (a library call: vmc_pde_amd/rocsolver.py), W = Q Z by
        `vmcpde_apply_reflectors`.  Nothing synchronises here: stedc's status word is copied to the host behind the call and is part of the
        step's health record -- `TDVP._check_guard` (run by `check()`, by `__call__` with observables, and at the top of the next right-hand
        side) rejects the step with FloatingPointError when it is non-zero and switches the core to the library solver, as
        torch.linalg.eigh raises on a non-zero info.  The returned W is backed by the workspace td["Z"] (overwritten by the next call)."""
        from . import rocsolver
        st = getattr(self, "_tridiag", None)
        if st is None:
            st = self._tridiag = dict(off=False, event=None, host=torch.zeros(1, dtype=torch.int32).pin_memory())
        # (sized for what THIS core needs: the 256-frame of the one-workgroup kernel, or -- ranks 257 .. 384, BASELINE C3 -- the frame of
        # vmcpde_sytrd_medium with the workspace of its leading r - 256 two-launch steps)
        nmax = int(nat.lib().vmcpde_sytrd_small_max_n()) if r <= int(nat.lib().vmcpde_sytrd_small_max_n()) else int(nat.lib().vmcpde_sytrd_medium_max_n())
        td = w.get("td")
        if td is None or td["nmax"] < nmax:
            nwork = max(int(nat.lib().vmcpde_sytrd_medium_workspace_bytes(nmax)) // 8, 1)
            td = w["td"] = dict(nmax=nmax, D=torch.empty(nmax, dtype=global_defs.tReal, device=device), E=torch.empty(nmax, dtype=global_defs.tReal, device=device),
                                tau=torch.empty(nmax, dtype=global_defs.tReal, device=device),
                                V=torch.empty((nmax, nmax), dtype=global_defs.tReal, device=device),
                                Z=torch.empty(nmax * nmax, dtype=global_defs.tReal, device=device),
                                work=torch.empty(nwork, dtype=global_defs.tReal, device=device),
                                info=torch.zeros(1, dtype=torch.int32, device=device))
        nmax = td["nmax"]
        Tc = T if T.is_contiguous() else T.contiguous()
        nat.call("vmcpde_sytrd_medium", nat.ptr(Tc), Tc.stride(0), r, nat.ptr(td["D"]), nat.ptr(td["E"]), nat.ptr(td["tau"]), nat.ptr(td["V"]), nmax,
                 nat.ptr(td["work"]), td["work"].numel() * 8, nat.stream())
        Z = td["Z"][:r * r].view(r, r)
        rocsolver.stedc(td["D"][:r], td["E"][:r], Z, td["info"])
        st["host"].copy_(td["info"], non_blocking=True)
        st["event"] = torch.cuda.Event()
        st["event"].record()
        st["pending"] = True                     # part of THIS step's acceptance: TDVP._check_guard reads the word with the solve's health record
        nat.call("vmcpde_apply_reflectors", nat.ptr(td["V"]), nmax, nat.ptr(td["tau"]), r, nat.ptr(Z), r, r, nat.stream())
        # (W is a view of the reused workspace td["Z"]: valid until the next call of this method -- `_eigh_lowrank` consumes it at once, U = Q1 W)
        return td["D"][:r].clone(), Z.t()

    # ---- the r x r core: one-sided Jacobi on the triangular factor (csrc/jacobi.hip) ------------------------------------
    def _core_uses_jacobi(self, r):
        """coreEigensolver="jacobi" (or VMCPDE_CORE_EIGH=jacobi; the default is the library's syevd, which is faster) for ranks the kernel
        takes well (<= 512), unless an earlier call's sweeps did not converge (read one call later, like every other health record of the
        engine: nothing synchronises inside a call)."""
        want = os.environ.get("VMCPDE_CORE_EIGH") or getattr(self, "coreEigensolver", "library")
        if want != "jacobi" or r > 512:        # (beyond 512 rows a workgroup's LDS holds 8 columns a side: tens of sweeps)
            return False
        st = getattr(self, "_jacobi", None)
        if st is not None and st["event"] is not None and not st["off"]:
            st["event"].synchronize()
            n = st["launched"]
            done = [s for s in range(n) if int(st["host"][s]) == 0]
            if not done:
                import warnings
                warnings.warn(f"Jacobi core of the spectral solve: {n} sweeps did not converge (rotations per sweep {st['host'][:n].tolist()}); "
                              f"using the library eigensolver from now on", RuntimeWarning)
                st["off"] = True
            else:
                st["n_sweeps"] = min(max(done[0] + 2, 4), 20)      # what the last call needed (its confirming sweep included) and one to spare
            st["event"] = None
        return st is None or not st["off"]

    def _core_eigh_jacobi(self, w, T, R1t, r, device):
        """(wv ascending, W) with T W = W diag(wv) for T = Q1^T S Q1 = R1 R1^T + Q1^T R Q1.  R1t = (Q1^T L)^T, r x r: row k = column k of R1.
        The columns of R1 are orthogonalised by Jacobi rotations (left singular vectors u_j, singular values sigma_j); the eigenvalues are
        then taken as Rayleigh quotients u_j^T T u_j, which folds the remainder R of the factorisation in (|Q1^T R Q1| <= trace(R): the
        size of a dense eigensolver's backward error) and costs two r x r products."""
        st = getattr(self, "_jacobi", None)
        if st is None:
            st = self._jacobi = dict(n_sweeps=20, launched=0, off=False, event=None, host=torch.zeros(32, dtype=torch.int32).pin_memory())
        rows, cols = nat.C.c_int32(), nat.C.c_int32()
        nat.call("vmcpde_jacobi_shape", r, nat.C.byref(rows), nat.C.byref(cols))
        rows, cols = rows.value, cols.value
        if w.get("jc") is None or w["jc"].shape != (cols, rows):
            w["jc"] = torch.zeros((cols, rows), dtype=global_defs.tReal, device=device)
            w["jc_counters"] = torch.zeros(32, dtype=torch.int32, device=device)
        CT = w["jc"]
        CT.zero_()
        CT[:r, :r] = R1t
        n = st["n_sweeps"]
        nat.call("vmcpde_jacobi_columns", nat.ptr(CT), r, n, nat.ptr(w["jc_counters"]), nat.stream())
        st["host"].copy_(w["jc_counters"], non_blocking=True)
        st["launched"] = n
        st["event"] = torch.cuda.Event()
        st["event"].record()
        sig2 = (CT * CT).sum(1)
        idx = torch.sort(sig2).indices[cols - r:]                  # (the zero padding sorts first)
        W = (CT[idx, :r] / torch.sqrt(sig2[idx])[:, None]).T.contiguous()
        wv = ((T @ W) * W).sum(0)
        order = torch.argsort(wv)
        return wv[order], W[:, order].contiguous()

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


    # ---- what the routes inherit from the previous call (for TDVP.state_dict) -----------------------------------------
    def _lowrank_state_dict(self):
        """The static plan of the Cholesky route (rmax, block steps, refinements, the previous call's rank / done / step count) and the
        pivot list with its length -- the candidates of the next factorisation (`vmcpde_lowrank_hint`).  With them the first call of a
        resumed run IS the next static call of the interrupted one, bit for bit."""
        out = {"last_rank": int(self._lowrank_last_rank), "chol_backoff": int(self._chol_backoff), "lowrank_backoff": int(self._lowrank_backoff),
               "gate_svdtol": getattr(self, "_lowrank_gate_svdtol", None), "gate_retry": int(getattr(self, "_lowrank_gate_retry", 0))}
        w = next(iter(self._buf_lr.values())) if getattr(self, "_buf_lr", None) else None
        if w is not None:
            torch.cuda.current_stream(w["piv"].device).synchronize()         # (the asynchronous host copies of the last call have landed)
            out["factor"] = {"P": int(next(iter(self._buf_lr))[1]), "piv": w["piv"].detach().cpu().clone(), "state": w["state"].detach().cpu().clone(),
                             "head": w["head_host"].clone()}
        st = self._chol_static
        if st is not None:
            if st["event"] is not None:
                st["event"].synchronize()
            out["chol_static"] = {k: st[k] for k in ("P", "rmax", "nsteps", "nref", "rank")}
            out["chol_static"].update(host=st["host"].clone(), head=st["head"].clone(), called=st["event"] is not None,
                                      frozen_run=int(st.get("frozen_run", 0)), frozen_last=bool(st.get("frozen_last", False)),
                                      finfo=None if st.get("finfo") is None else st["finfo"].clone())
        return out

    def _lowrank_load_state(self, st):
        self._lowrank_last_rank = int(st.get("last_rank", 0))
        self._chol_backoff, self._lowrank_backoff = int(st.get("chol_backoff", 0)), int(st.get("lowrank_backoff", 0))
        self._lowrank_gate_svdtol, self._lowrank_gate_retry = st.get("gate_svdtol"), int(st.get("gate_retry", 0))
        self._buf_lr = {}
        self._lr_restore = st.get("factor")
        plan = st.get("chol_static")
        if self._lr_restore is not None and plan is not None and plan.get("called"):
            self._lr_restore = dict(self._lr_restore, ws_head=plan["head"].clone())        # (what the last static call left in the workspace head)
        self._chol_static = None
        if plan is not None:
            ev = None
            if plan["called"]:                      # (the record of the interrupted run's last static call is already on the host)
                ev = torch.cuda.Event()
                ev.record()
            self._chol_static = dict(P=plan["P"], rmax=plan["rmax"], nsteps=plan["nsteps"], nref=plan["nref"], rank=plan["rank"], event=ev,
                                     host=plan["host"].clone().pin_memory(), head=plan["head"].clone().pin_memory(),
                                     frozen_run=int(plan.get("frozen_run", 0)), frozen_last=bool(plan.get("frozen_last", False)),
                                     finfo=None if plan.get("finfo") is None else plan["finfo"].clone())
