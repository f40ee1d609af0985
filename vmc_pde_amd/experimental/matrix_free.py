"""EXPERIMENT (opt-in, never the headline): a TDVP step without an assembled S.

The pivoted low-rank factor of S = <J^T J> (lowrank.py) touches S only through its diagonal and the columns it pivots on, and the
refinement and the diagnostics only through products S0 v.  All of them are formed from the retained Jacobian: for the previous call's
pivots AHEAD in one pass (pivot hints), 32 at a time on demand otherwise (csrc/eig.hip::colpanel_kernel), S0 v = J^T (J v) / N - m (m.v)
(vmcpde_jt_matvec) -- ~7 passes over J and a 224-column GEMM instead of N P^2 flop.  north_star prescribes an assembled S and an MFMA
syrk; a full-rank S (BASELINE config 4) has no shortcut here and raises.  Reported beside the headline.
"""
from __future__ import annotations

import os

import torch

from .. import _native as nat
from .. import global_defs
from .. import mpi_wrapper as mpi
from ..lowrank import LOWRANK_RANK_CAP, LOWRANK_RANK_DIVISOR, LOWRANK_TOL, StaticPlanOutdated, _round_up


class MatrixFreeMixin:
    """Overrides `solve`, `_drop_plans` and the attribute `S0` of `tdvp.TDVP`; state: `_mf_plan` (the static plan: rmax, block steps, the
    previous call's rank / step count read one call later), `_S0_lazy`."""

    def _init_matrix_free(self):
        self._mf_plan = None
        self._S0, self._S0_lazy = None, False

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

    def _drop_plans(self):
        super()._drop_plans()
        self._mf_plan = None

    def solve(self, Eloc, gradients, logProbs, psi=None):
        if self.matrixFree:
            return self._solve_matrix_free(Eloc, gradients, logProbs, psi)
        return super().solve(Eloc, gradients, logProbs, psi=psi)

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
                    # the record queued before the extra steps says "not finished": replace it, or the next call would discard the plan
                    # (and the enlarged step count with it) and plan anew every time
                    self._mf_plan["nsteps"] += extra
                    if mode == "static":
                        plan["host"].copy_(w["state"], non_blocking=True)
                        plan["head"].copy_(w["ws"][:5], non_blocking=True)
                        plan["event"].record()
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
                Tm = 0.5 * (Tm + Tm.T)
                if self._core_choice(r) == "tridiag":               # (the product's own core: csrc/tridiag.hip + stedc, lowrank.py)
                    wv, W = self._core_eigh_tridiag(w, Tm, r, device)
                else:
                    wv, W = torch.linalg.eigh(Tm)
                self.ev, self.V = self._lowrank_assemble(Y, Xn, Q1 @ W, wv, r, P, device)
                self._ritz = None
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

