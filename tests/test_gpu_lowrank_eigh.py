"""The rank-revealing spectral decomposition (csrc/eig.hip + TDVP._eigh_lowrank; reference: tdvp.py:57-64, eigh of the full matrix):
its two dependent chains against numpy restatements, the assembled (ev, V) against a dense eigh on the same S at the headline network
size, and when the engine is (not) taken."""
import numpy as np
import pytest
import torch

from oracle import vmc_oracle as O
from vmc_pde_amd.experimental import TDVP as XTDVP      # the two opt-in experiments (matrixFree, int8Gram) live in a subclass
from helpers import make_spec, random_theta, make_varstate, FixedSampler

pytestmark = pytest.mark.gpu


def _graded_psd(P, per_decade, N, seed):
    """S = J^T J / N for J with singular values falling by a factor ten every `per_decade` directions (what these models produce)."""
    rng = np.random.default_rng(seed)
    lam = 3.0e3 * 10.0 ** (-np.arange(P) / per_decade)
    Q, _ = np.linalg.qr(rng.standard_normal((P, P)))
    J = rng.standard_normal((N, P)) * np.sqrt(lam) @ Q.T
    return J.T @ J / N


def _pivoted_cholesky_numpy(S, tol_rel):
    P = S.shape[0]
    d = np.diag(S).copy()
    d0 = d.max()
    cols, piv = [], []
    while len(piv) < P:
        p = int(np.argmax(d))
        if not d[p] > tol_rel * d0:
            break
        L = np.array(cols).T if cols else np.zeros((P, 0))
        col = (S[:, p] - L @ L[p]) / np.sqrt(d[p])
        col[piv] = 0.0
        cols.append(col)
        d = d - col * col
        piv.append(p)
        d[piv] = -np.inf
    return np.array(cols).T, piv


def _run_pivchol(S, tol, cap, chunks):
    from vmc_pde_amd import _native as nat
    P = S.shape[0]
    Sd = torch.from_numpy(S).cuda()
    ldl = (P + 63) // 64 * 64
    LT = torch.full((cap, ldl), float("nan"), dtype=torch.float64, device="cuda")
    piv = torch.zeros(cap, dtype=torch.int32, device="cuda")
    state = torch.zeros(2, dtype=torch.int32, device="cuda")
    ws = torch.zeros(nat.lib().vmcpde_pivoted_cholesky_workspace_bytes(P) // 8, dtype=torch.float64, device="cuda")
    step = 0
    for n in chunks:
        n = min(n, cap - step)
        nat.call("vmcpde_pivoted_cholesky", nat.ptr(Sd), P, P, tol, step, n, cap, nat.ptr(LT), ldl, nat.ptr(piv), nat.ptr(state), nat.ptr(ws),
                 ws.numel() * 8, nat.stream())
        step += n
    torch.cuda.synchronize()
    r, done = int(state[0]), int(state[1])
    return LT[:r].cpu().numpy(), piv[:r].cpu().numpy(), r, done, ws[:2].cpu().numpy()


def test_pivoted_cholesky_against_numpy():
    P = 700                                                       # ragged: 700 = 10 * 64 + 60
    S = _graded_psd(P, 9.0, 4096, seed=5)
    L_ref, piv_ref = _pivoted_cholesky_numpy(S, 1e-16)
    LT, piv, r, done, head = _run_pivchol(S, 1e-16, cap=400, chunks=(64, 1, 100, 235))      # resumed three times
    assert done == 1 and abs(r - len(piv_ref)) <= 3, (r, len(piv_ref))
    # the leading pivots are well separated: same choice, same columns (later ones are decided by round-off)
    m = r // 2
    assert list(piv[:m]) == piv_ref[:m]
    assert np.abs(LT[:m, :P].T - L_ref[:, :m]).max() < 1e-9 * np.abs(L_ref).max()
    # rows keep their place; a row chosen as pivot has structural zeros in every later column, and pad columns are zero
    for k in range(1, r):
        assert np.all(LT[k, piv[:k]] == 0.0)
    assert np.all(LT[:, P:] == 0.0) and len(set(piv.tolist())) == r
    # S = L L^T + R with trace(R) as reported: R is what the stopping criterion left
    lmax = np.linalg.eigvalsh(S)[-1]
    R = S - LT[:, :P].T @ LT[:, :P]
    assert np.abs(R).max() < 1e-15 * lmax * 10
    assert head[0] == np.diag(S).max() and 0.0 <= head[1] <= (P - r) * 1e-16 * head[0]
    # a cap below the rank: the loop ends without the criterion, state says so, and the trace of what is left is still reported
    LT2, piv2, r2, done2, head2 = _run_pivchol(S, 1e-16, cap=40, chunks=(40,))
    assert done2 == 0 and r2 == 40 and list(piv2) == piv_ref[:40]
    left = np.trace(S - LT2[:, :P].T @ LT2[:, :P])
    assert abs(head2[1] - left) < 1e-9 * abs(left)
    # a matrix with a NaN, and a zero matrix: rank 0, criterion "met" at the first step
    for bad in (np.full((70, 70), np.nan), np.zeros((70, 70))):
        _, _, r3, done3, _ = _run_pivchol(bad, 1e-16, cap=10, chunks=(10,))
        assert r3 == 0 and done3 == 1


def _run_pivchol_blocked(S, tol, cap, calls):
    from vmc_pde_amd import _native as nat
    P = S.shape[0]
    Sd = torch.from_numpy(S).cuda()
    ldl = (P + 63) // 64 * 64
    LT = torch.full((cap, ldl), float("nan"), dtype=torch.float64, device="cuda")
    piv = torch.zeros(cap, dtype=torch.int32, device="cuda")
    state = torch.zeros(2, dtype=torch.int32, device="cuda")
    ws = torch.zeros(nat.lib().vmcpde_pivoted_cholesky_blocked_workspace_bytes(P) // 8, dtype=torch.float64, device="cuda")
    for c, nb in enumerate(calls):
        nat.call("vmcpde_pivoted_cholesky_blocked", nat.ptr(Sd), P, P, tol, 1 if c == 0 else 0, nb, cap, nat.ptr(LT), ldl, nat.ptr(piv), nat.ptr(state),
                 nat.ptr(ws), ws.numel() * 8, nat.stream())
    torch.cuda.synchronize()
    r, done = int(state[0]), int(state[1])
    return LT[:r].cpu().numpy(), piv[:r].cpu().numpy(), r, done, ws[:2].cpu().numpy()


@pytest.mark.parametrize("P,per_decade", [(700, 9.0), (4213, 14.0), (130, 2.5)])
def test_blocked_pivoted_cholesky(P, per_decade):
    """Up to 32 pivots per pair of launches, in whatever order the candidates come: the properties that make it a pivoted Cholesky
    factorisation, and the same numerical rank as the greedy order."""
    S = _graded_psd(P, per_decade, 2048 if P > 2000 else 4096, seed=8)
    _, piv_ref = _pivoted_cholesky_numpy(S, 1e-14) if P <= 1000 else (None, None)
    cap = min(P, 600)
    LT, piv, r, done, head = _run_pivchol_blocked(S, 1e-14, cap, calls=(2, 1, 40))        # resumed twice
    assert done == 1 and 0 < r < cap
    if piv_ref is not None:
        assert abs(r - len(piv_ref)) <= max(4, len(piv_ref) // 20), (r, len(piv_ref))
    assert len(set(piv.tolist())) == r and piv.min() >= 0 and piv.max() < P
    L = LT[:, :P].T                                                    # P x r
    assert np.all(np.isfinite(LT)) and np.all(LT[:, P:] == 0.0)
    # a pivot row has a positive entry in its own column and structural zeros in every later one
    for k in range(r):
        assert L[piv[k], k] > 0.0
        assert np.all(L[piv[k], k + 1:] == 0.0)
    # S = L L^T + R, R what the criterion left: every remaining diagonal entry <= 1e-14 max diag (up to the round-off of the downdates)
    d0 = np.diag(S).max()
    R = S - L @ L.T
    assert head[0] == d0
    assert np.diag(R).max() < 1.2e-14 * d0 and np.abs(R).max() < 1.2e-14 * d0 and np.diag(R).min() > -1e-14 * d0, (np.diag(R).max() / d0, np.diag(R).min() / d0)
    assert 0.0 <= head[1] <= (P - r) * 1e-14 * d0
    # the factor spans the range of S as well as the greedy one: eigenvalues of L L^T against those of S
    ev = np.linalg.eigvalsh(S)
    ev_l = np.linalg.eigvalsh(L.T @ L)
    assert np.abs(ev[-r:] - ev_l).max() < 1e-14 * ev[-1]
    # the rank cap: reached without the criterion
    cap2 = min(40, r // 2)
    LT2, piv2, r2, done2, _ = _run_pivchol_blocked(S, 1e-14, cap=cap2, calls=(20,))       # (130 rows are 3 row blocks: 3 candidates per step)
    assert done2 == 0 and r2 == cap2 and len(set(piv2.tolist())) == cap2
    assert np.abs(S - LT2[:, :P].T @ LT2[:, :P]).max() < np.abs(S).max()          # (a partial factorisation: positive semi-definite remainder)
    assert np.linalg.eigvalsh(S - LT2[:, :P].T @ LT2[:, :P]).min() > -1e-13 * d0


def test_pivot_hints_change_the_candidates_and_nothing_else():
    """vmcpde_lowrank_hint: the pivots of a finished factorisation as preferred candidates of the next one, on a nearby matrix -- the
    same properties of the result in fewer block steps; a useless or hostile list (pivots of an unrelated matrix, rows out of range,
    repeats) costs steps, never validity; disarmed, the factorisation is bit for bit the unhinted one."""
    from vmc_pde_amd import _native as nat
    P, cap = 1500, 600
    S1 = _graded_psd(P, 12.0, 4096, seed=21)
    rng = np.random.default_rng(22)
    E = rng.standard_normal((P, 40)) * np.sqrt(1e-8 * np.diag(S1).max())
    S2 = S1 + E @ E.T                                                      # a nearby PSD matrix: a rank-40 term at 1e-8 of the scale
    S3 = _graded_psd(P, 12.0, 4096, seed=99)                                                                 # an unrelated one
    ldl = (P + 63) // 64 * 64
    LT = torch.zeros((cap, ldl), dtype=torch.float64, device="cuda")
    piv = torch.zeros(cap, dtype=torch.int32, device="cuda")
    state = torch.zeros(2, dtype=torch.int32, device="cuda")
    wsb = nat.lib().vmcpde_pivoted_cholesky_blocked_workspace_bytes(P)
    ws = torch.zeros(wsb // 8, dtype=torch.float64, device="cuda")

    def factor(S, hint=None):
        """hint: None = leave the workspace's list as it is; False = disarm; (piv, state) tensors = arm with them"""
        Sd = torch.from_numpy(S).cuda()
        if hint is False:
            nat.call("vmcpde_lowrank_hint", None, None, P, nat.ptr(ws), wsb, nat.stream())
        elif hint is not None:
            nat.call("vmcpde_lowrank_hint", nat.ptr(hint[0]), nat.ptr(hint[1]), P, nat.ptr(ws), wsb, nat.stream())
        LT.zero_()
        nat.call("vmcpde_pivoted_cholesky_blocked", nat.ptr(Sd), P, P, 1e-14, 1, 60, cap, nat.ptr(LT), ldl, nat.ptr(piv), nat.ptr(state), nat.ptr(ws),
                 wsb, nat.stream())
        torch.cuda.synchronize()
        r, done, steps = int(state[0]), int(state[1]), int(ws[:4].view(torch.int32)[7])
        L = LT[:r, :P].cpu().numpy().T
        pv = piv[:r].cpu().numpy()
        d0 = np.diag(S).max()
        R = S - L @ L.T
        assert done == 1 and 0 < r < cap and len(set(pv.tolist())) == r and pv.min() >= 0 and pv.max() < P
        assert np.diag(R).max() < 1.2e-14 * d0 and np.abs(R).max() < 1.2e-14 * d0 and np.diag(R).min() > -1e-14 * d0
        for k in range(r):
            assert L[pv[k], k] > 0.0 and np.all(L[pv[k], k + 1:] == 0.0)
        return r, steps, LT[:r].clone(), piv[:r].clone()

    r1, steps1, LT1, piv1 = factor(S1, hint=False)
    assert steps1 >= r1 // 20
    # (these random-basis matrices have no correlated neighbours that collapse inside a block, so the unhinted yield is high already --
    # the step count of the model's own S is asserted in test_lowrank_cholesky_solve_matches_the_dense_factorisation)
    r1h, steps1h, _, _ = factor(S1, hint=(piv, state))                    # its own pivots
    assert abs(r1h - r1) <= max(4, r1 // 20) and steps1h <= steps1, (steps1, steps1h)
    r2h, steps2h, _, _ = factor(S2, hint=(piv, state))                    # the nearby matrix with the list of the last factorisation
    r2, steps2, _, _ = factor(S2, hint=False)
    assert abs(r2h - r2) <= max(4, r2 // 20) and steps2h <= steps2, (steps2, steps2h)
    # the list of an unrelated matrix: valid result (checked inside), never more than a few steps beyond the unhinted count
    r3, steps3, _, _ = factor(S3, hint=False)
    factor(S1, hint=False)
    r3h, steps3h, _, _ = factor(S3, hint=(piv, state))
    assert abs(r3h - r3) <= max(4, r3 // 20) and steps3h <= steps3 + max(4, steps3 // 2), (steps3, steps3h)
    # a hostile list: rows out of range, negative, repeated
    bad = torch.tensor([5, 5, 5, -3, P, P + 7, 2 ** 30, 17, 17, 0] * 20, dtype=torch.int32, device="cuda")
    bad_state = torch.tensor([bad.numel(), 1], dtype=torch.int32, device="cuda")
    rb, stepsb, _, _ = factor(S1, hint=(bad, bad_state))
    assert abs(rb - r1) <= max(4, r1 // 20)
    # disarmed: bit for bit the first factorisation
    r1b, steps1b, LT1b, piv1b = factor(S1, hint=False)
    assert r1b == r1 and steps1b == steps1 and torch.equal(piv1b, piv1) and torch.equal(LT1b, LT1)
    # an uninitialised workspace carries no hint (the armed flag is a magic word)
    ws.view(torch.int32)[4:8] = torch.tensor([12345, 1 << 20, 7, 3], dtype=torch.int32, device="cuda")
    r1c, _, LT1c, piv1c = factor(S1)
    assert r1c == r1 and torch.equal(piv1c, piv1) and torch.equal(LT1c, LT1)


@pytest.mark.parametrize("P,r", [(613, 90), (613, 91), (5000, 7), (1500, 2), (700, 1), (4097, 4), (9000, 5), (17000, 9)])
@pytest.mark.parametrize("block", [2, 4])
def test_householder_rows_against_numpy(P, r, block, monkeypatch):
    """(two reflectors per launch up to 16 384 rows -- four with VMCPDE_HQR_BLOCK=4 up to 8192, then two for what is left --, a
    single-step launch for the odd one out and beyond)"""
    from vmc_pde_amd import _native as nat
    monkeypatch.setenv("VMCPDE_HQR_BLOCK", str(block))
    rng = np.random.default_rng(2)
    L = rng.standard_normal((P, r)) * 10.0 ** (-np.arange(r) / 11.0)              # columns over eight decades
    if r > 17:
        L[:, 17] = 0.0
        L[17, 17] = 0.3                                                            # a column that is (after 17 steps: nearly) a unit vector
    ldl = (P + 63) // 64 * 64
    LT = torch.zeros((r, ldl), dtype=torch.float64, device="cuda")
    LT[:, :P] = torch.from_numpy(L.T)
    YT = torch.full((r, ldl), float("nan"), dtype=torch.float64, device="cuda")
    tau = torch.zeros(r, dtype=torch.float64, device="cuda")
    scratch = torch.zeros(5 * r + 5, dtype=torch.float64, device="cuda")
    nat.call("vmcpde_householder_rows", nat.ptr(LT), ldl, P, r, nat.ptr(YT), nat.ptr(tau), nat.ptr(scratch), nat.stream())
    torch.cuda.synchronize()
    Y, t, Rm = YT.cpu().numpy()[:, :P].T, tau.cpu().numpy(), LT.cpu().numpy()[:, :P].T
    assert np.all(YT.cpu().numpy()[:, P:] == 0.0) and np.all(np.triu(Y[:r], 1) == 0.0) and np.all(np.diag(Y[:r]) == 1.0)
    assert np.all(t > 0.0) and np.all(t <= 2.0)
    # compact WY: Q = I - Y T Y^T with T^-1 = strict upper(Y^T Y) + diag(1 / tau)
    Tinv = np.triu(Y.T @ Y, 1) + np.diag(1.0 / t)
    if P <= 2000:
        Q = np.eye(P) - Y @ np.linalg.solve(Tinv, Y.T)
        assert np.abs(Q.T @ Q - np.eye(P)).max() < 5e-15
        QtL = Q.T @ L
    else:                                       # (large P: Q implicitly -- its first r columns, what the engine uses, and Q^T L)
        Q1 = -Y @ np.linalg.solve(Tinv, Y[:r].T)
        Q1[:r] += np.eye(r)
        assert np.abs(Q1.T @ Q1 - np.eye(r)).max() < 5e-15
        QtL = L - Y @ np.linalg.solve(Tinv.T, Y.T @ L)
    # Q^T L = [R; 0], column by column relative to the column's norm (the factor's columns differ by 1e8 in size)
    cn = np.linalg.norm(L, axis=0)
    assert np.abs(QtL[r:] / cn).max() < 1e-14
    # what the kernel leaves in LT: the strict upper part of R (column k itself is never rewritten -- every workgroup of step k reads it)
    assert np.abs((np.triu(QtL[:r], 1) - np.triu(Rm[:r], 1)) / cn).max() < 1e-13
    # and the same R as LAPACK up to the sign of each row
    R_np = np.linalg.qr(L, mode="r")
    assert np.abs((np.abs(np.triu(QtL[:r])) - np.abs(R_np)) / cn).max() < 1e-12


@pytest.fixture(scope="module")
def headline_net():
    """The headline network (d = 2, 4 blocks, hidden (21, 21): P = 4213) with injected latent samples, as a factory of VarStates."""
    import vmc_pde_amd as V
    spec = make_spec(2, 4, (21, 21), latent="Gauss", coupling="no_add", offset=[1.0, 1.0], seed=4)
    th = random_theta(spec, 4, out_scale=0.03)
    ep = V.EvolutionEquation(dim=2, name="advection_hamiltonian_wDiss")
    ep.eqParams["advection_hamiltonian_wDiss"]["params"]["lam"] = 0.1

    def varstate(N):
        g = torch.Generator().manual_seed(9)
        z = O.latent_from_normals(spec, th, torch.randn(N, 2, generator=g, dtype=torch.float64), None)
        return make_varstate(spec, th, sampler=FixedSampler(z.cuda()))
    return varstate, ep, th


def test_lowrank_engine_matches_dense_at_headline_size(headline_net):
    import vmc_pde_amd as V
    varstate, ep, th = headline_net
    N = 8192
    vs = varstate(N)
    out = {}
    for engine in ("dense", "lowrank", "auto"):
        tp = V.TDVP(svdTol=1e-6, spectralEngine=engine)
        u, _ = tp(th.cuda(), 0.0, vs, ep, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
        out[engine] = (tp.S0.clone(), tp.F0.clone(), tp.ev.clone(), tp.V.clone(), tp.VtF.clone(), u.clone(), dict(tp.spectralInfo),
                       float(tp.solverResidual), float(tp.tdvp_error))
    S, F, ev_d, V_d, _, u_d, info_d, res_d, err_d = out["dense"]
    P = S.shape[0]
    assert P == 4213 and info_d["engine"] == "dense"
    for engine in ("lowrank", "auto"):
        S2, F2, ev, Vv, VtF, u, info, res, err = out[engine]
        assert torch.equal(S2, S) and torch.equal(F2, F)                 # same samples, same S: what differs is the decomposition
        assert info["engine"] == "lowrank" and info["converged"] and info["rank"] < P // 6, info
        lmax = float(ev_d[-1])
        assert float((ev - ev_d).abs().max()) < 1e-13 * lmax
        assert bool((ev[1:] >= ev[:-1]).all()) and int((ev == 0).sum()) == P - info["rank"]
        eye = torch.eye(P, dtype=torch.float64, device="cuda")
        assert float((Vv.T @ Vv - eye).abs().max()) < 1e-13
        # residual of the decomposition against the dense one's own
        r_lr = float(torch.linalg.norm(S @ Vv - Vv * ev)) / lmax
        r_d = float(torch.linalg.norm(S @ V_d - V_d * ev_d)) / lmax
        assert r_lr < max(3 * r_d, 1e-13), (r_lr, r_d)
        # the solve: update in the S-norm, the reference's two diagnostics, and the invariants of every eigenvalue cluster above 1e-9
        nF = float(torch.linalg.norm(F))
        assert float(torch.linalg.norm(S @ (u - u_d))) < 1e-10 * nF
        assert abs(res - res_d) < 1e-10 and abs(err - err_d) < 1e-10
        evc = ev_d.cpu()
        bounds = [0] + [k for k in range(1, P) if float(evc[k] - evc[k - 1]) >= 1e-6 * lmax] + [P]
        f_lr, f_d = (VtF ** 2).cpu(), ((V_d.T @ F) ** 2).cpu()
        n = 0
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            if float(evc[hi - 1]) < 1e-9 * lmax:
                continue
            n += 1
            assert abs(float(f_lr[lo:hi].sum()) - float(f_d[lo:hi].sum())) < 1e-8 * float(f_d[lo:hi].sum())
        assert n >= 20


def test_engine_choice_and_fallbacks(headline_net):
    import vmc_pde_amd as V
    varstate, ep, th = headline_net
    N = 4096
    vs = varstate(N)
    kw = dict(nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
    # a diagonal shift makes S full rank: the dense path is taken without an attempt
    tp = V.TDVP(svdTol=1e-6, diagonalShift=1e-3)
    tp(th.cuda(), 0.0, vs, ep, **kw)
    assert tp.spectralInfo == dict(engine="dense")
    # a full-rank matrix handed to the engine under "auto": the cap is hit, the dense result is returned and the engine pauses
    tq = V.TDVP(svdTol=1e-6)
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn(3000, 1200, generator=g, dtype=torch.float64, device="cuda")
    S = (A.T @ A / 3000).contiguous()
    F = torch.randn(1200, generator=g, dtype=torch.float64, device="cuda")
    tq.transform_to_eigenbasis(S, F)
    assert tq.spectralInfo["engine"] == "dense" and tq.spectralInfo["tried"] == "lowrank" and not tq.spectralInfo["converged"]
    assert tq.spectralInfo["rank"] == 1200 // 6 and tq._lowrank_backoff > 0
    ev_ref = torch.linalg.eigvalsh(S)
    assert float((tq.ev - ev_ref).abs().max()) < 1e-12 * float(ev_ref[-1])
    before = tq._lowrank_backoff
    tq.transform_to_eigenbasis(S, F)                                  # no second attempt right away
    assert tq.spectralInfo == dict(engine="dense") and tq._lowrank_backoff == before - 1
    # forced: no cap, the factorisation runs to full rank and the result is still a decomposition of S
    tf = V.TDVP(svdTol=1e-6, spectralEngine="lowrank")
    tf.transform_to_eigenbasis(S, F)
    assert tf.spectralInfo["engine"] == "lowrank" and tf.spectralInfo["rank"] == 1200
    assert float((tf.ev - ev_ref).abs().max()) < 1e-12 * float(ev_ref[-1])
    assert float(torch.linalg.norm(S @ tf.V - tf.V * tf.ev)) < 1e-12 * float(ev_ref[-1])
    with pytest.raises(ValueError, match="spectralEngine"):
        V.TDVP(spectralEngine="jacobi")
    # a NaN in S is not "a smaller pivot": the engine declines and the dense path reports it as before
    Sn = S.clone()
    Sn[17, 17] = float("nan")
    assert V.TDVP(spectralEngine="lowrank")._eigh_lowrank(Sn) is None
    low = (A[:, :40] @ A[:, :40].T / 40)[:1200, :1200].contiguous()             # rank 40: the engine applies ...
    tl = V.TDVP()
    assert tl._eigh_lowrank(low) is not None and tl.spectralInfo["rank"] == 40
    low[5, 5] = float("inf")                                                      # ... and declines once an entry is not finite
    assert tl._eigh_lowrank(low) is None


def test_static_calls_on_the_frozen_pivot_set(headline_net):
    """Round 6: a call on a static plan factors S on the PREVIOUS factorisation's pivot set without searching (gather, one tall Cholesky, transpose)
    -- the same regularised system solved to the same accuracy (fresh samples every call: S moves by sampling noise), the remainder measured and
    reported one call later, a pivoted factorisation every FROZEN_REFRESH_AFTER + 1 calls, and a set that no longer spans S sends the next call back
    to the pivoted factorisation while its own solve is still refined against the true matrix and guarded by the residual."""
    import vmc_pde_amd as V
    import vmc_pde_amd.lowrank as LR
    from helpers import rel_err
    _, ep, _ = headline_net
    spec = make_spec(2, 4, (21, 21), latent="Gauss", coupling="no_add", offset=[1.0, 1.0], seed=4)
    th = random_theta(spec, 4, out_scale=0.03)
    N = 8192
    vs, vs_d = make_varstate(spec, th), make_varstate(spec, th)            # the product's own sampler: fresh samples every call
    kw = dict(nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
    tp = V.TDVP(diagonalShift=1e-4, svdTol=1e-6, solver="cholesky")
    td = V.TDVP(diagonalShift=1e-4, svdTol=1e-6, solver="cholesky", choleskyEngine="dense")
    factors = []
    for call in range(LR.FROZEN_REFRESH_AFTER + 4):
        vs_d.sampler._calls = vs.sampler._calls                    # the twin draws the same samples
        u, _ = tp(th.cuda(), 0.0, vs, ep, **kw)
        tp.check()
        ud, _ = td(th.cuda(), 0.0, vs_d, ep, **kw)
        td.check()
        assert rel_err(td.S0, tp.S0) < 1e-12
        A = tp.S0 + torch.diag(1e-4 * torch.diag(tp.S0) + tp._floor)
        assert float(torch.linalg.norm(A @ (u - ud)) / torch.linalg.norm(tp.F0)) < 1e-10 and float(tp.solverResidual) < 1e-11, (call, tp.choleskyInfo)
        factors.append(tp.choleskyInfo.get("factor", tp.choleskyInfo["mode"]))
        if factors[-1] == "frozen pivots":
            assert tp.choleskyInfo["block_steps"] == 0
            tp._chol_static["event"].synchronize()
            rho, pinned, rmaxdiag = (float(v) for v in tp._chol_static["finfo"])
            st = tp._chol_static
            assert pinned == 0.0 and 0.0 <= rho < 1e-4 and int(st["host"][0]) == st["rank"] or int(st["host"][1]) == 1, (rho, pinned)
    n = LR.FROZEN_REFRESH_AFTER
    assert factors == ["planning"] + ["frozen pivots"] * n + ["pivoted"] + ["frozen pivots"] * 2, factors
    # a pivot set that does not span S (made so by hand: the first r parameters instead of the pivots): the call's own solve still refines against
    # the true matrix and its residual is judged by the guard; the measured remainder sends the NEXT call to the pivoted factorisation
    w = next(iter(tp._buf_lr.values()))
    r = tp._chol_static["rank"]
    w["piv"][:r] = torch.arange(r, dtype=torch.int32, device="cuda")
    tp(th.cuda(), 0.0, vs, ep, **kw)
    assert tp.choleskyInfo["factor"] == "frozen pivots"
    try:
        tp.check()
        ok = True
    except FloatingPointError:                                        # (StaticPlanOutdated: the plan is dropped, the next call plans anew)
        ok = False
    tp(th.cuda(), 0.0, vs, ep, **kw)
    tp.check()
    assert tp.choleskyInfo.get("factor", tp.choleskyInfo["mode"]) in ("pivoted", "planning") and (ok or tp.choleskyInfo["mode"] == "planning")
    assert float(tp.solverResidual) < 1e-11
    # the switch
    tn = V.TDVP(diagonalShift=1e-4, svdTol=1e-6, solver="cholesky", frozenPivots=False)
    for _ in range(3):
        tn(th.cuda(), 0.0, vs, ep, **kw)
    assert tn.choleskyInfo["factor"] == "pivoted"


def test_lowrank_cholesky_solve_matches_the_dense_factorisation(headline_net, monkeypatch):
    """solver="cholesky" through the pivoted factor + Woodbury + refinement against the blocked factorisation of the full matrix: the
    same regularised system, so the same update to round-off -- and the fallbacks when the low-rank route does not apply."""
    import vmc_pde_amd as V
    varstate, ep, th = headline_net
    N = 8192
    vs = varstate(N)
    kw = dict(nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
    out = {}
    import vmc_pde_amd.tdvp as T
    assert T.LOWRANK_CHOL_MIN_P <= 4213            # "auto" takes the low-rank route at the headline size
    for engine in ("dense", "lowrank", "auto"):
        tp = V.TDVP(diagonalShift=1e-4, svdTol=1e-6, solver="cholesky", choleskyEngine=engine)
        u, _ = tp(th.cuda(), 0.0, vs, ep, **kw)
        tp.check()
        out[engine] = (tp.S0.clone(), tp.F0.clone(), u.clone(), float(tp.solverResidual), float(tp.tdvp_error), dict(tp.choleskyInfo), tp)
    S, F, u_d, res_d, err_d, info_d, tp_d = out["dense"]
    assert info_d == dict(engine="dense")
    P = S.shape[0]
    D = 1e-4 * torch.diagonal(S) + 1e-6 * torch.diagonal(S).max()
    for engine in ("lowrank", "auto"):
        S2, F2, u, res, err, info, _ = out[engine]
        assert torch.equal(S2, S) and torch.equal(F2, F)
        assert info["engine"] == "lowrank" and info["converged"] and 1 <= info["rank"] < P // 6, info
        assert info["trace_left"] / info["min_shift"] < 1e-5 and abs(info["min_shift"] - float(D.min())) < 1e-12 * float(D.min())
        # the solution of the same system: against the dense factorisation and against the matrix itself
        # (the matrix has a condition number of ~3e7: two solutions with residuals of 1e-13 may differ by 1e-6 in norm; the statement is
        # about the residual of each and about their difference as the matrix sees it)
        nF = float(torch.linalg.norm(F))
        du = u - u_d
        assert float(torch.linalg.norm(S @ du + D * du)) < 1e-11 * nF, float(torch.linalg.norm(S @ du + D * du)) / nF
        assert float(torch.linalg.norm(S @ u + D * u - F)) < 1e-12 * nF and float(torch.linalg.norm(S @ u_d + D * u_d - F)) < 1e-11 * nF
        assert float(torch.linalg.norm(du) / torch.linalg.norm(u_d)) < 1e-4
        assert res < 1e-12 and res_d < 1e-12 and abs(err - err_d) < 1e-10
    # the calls after the first run in static mode (no host read): same system, same solution; the plan is checked one call later
    tp = out["lowrank"][6]
    assert tp.choleskyInfo["mode"] == "planning" and tp._chol_static is not None and tp._chol_static["rmax"] % 128 == 0
    tp.frozenPivots = False          # (this test is about the PIVOTED factorisation of a static call; test_static_calls_on_the_frozen_pivot_set has the other)
    steps_hinted = []
    steps_planned = tp._steps_used(tp._buf_lr[next(iter(tp._buf_lr))]["head_host"])      # the first call had no list
    for rep in range(3):
        u, _ = tp(th.cuda(), 0.0, vs, ep, **kw)
        tp.check()
        assert tp.choleskyInfo["mode"] == "static" and tp.choleskyInfo["rmax"] == tp._chol_static["rmax"], tp.choleskyInfo
        du = u - u_d
        assert float(torch.linalg.norm(S @ du + D * du)) < 1e-11 * nF and float(tp.solverResidual) < 1e-12
        if rep:
            assert int(tp._chol_static["host"][1]) == 1 and abs(int(tp._chol_static["host"][0]) - tp.choleskyInfo["planned_rank"]) <= 8
            # pivot hints: the previous call's pivots as candidates -- the same matrix here -- take about 2/3 of the unhinted block steps
            steps_hinted.append(tp._steps_used(tp._chol_static["head"]))
    assert steps_planned >= 12 and max(steps_hinted) <= (3 * steps_planned) // 4, (steps_planned, steps_hinted)
    # ... and without them (pivotHints=False) every call takes the unhinted count and returns the same solution
    tn = V.TDVP(diagonalShift=1e-4, svdTol=1e-6, solver="cholesky", choleskyEngine="lowrank", pivotHints=False, frozenPivots=False)
    for rep in range(3):
        un, _ = tn(th.cuda(), 0.0, vs, ep, **kw)
        tn.check()
    assert tn.choleskyInfo["mode"] == "static" and tn._steps_used(tn._chol_static["head"]) == steps_planned
    du = un - u_d
    assert float(torch.linalg.norm(S @ du + D * du)) < 1e-11 * nF and float(tn.solverResidual) < 1e-12
    # the plan is checked one call later: a rank within 32 columns of the pad makes the next call plan anew
    tp._chol_static["host"][0] = tp._chol_static["rmax"] - 20
    u, _ = tp(th.cuda(), 0.0, vs, ep, **kw)
    tp.check()
    assert tp.choleskyInfo["mode"] == "planning" and float(tp.solverResidual) < 1e-12
    # a plan far too small for the matrix (made so by hand: a quarter of the rank, four block steps): the call still returns, its
    # measured residual fails the guard LOUDLY at the next check, the plan is dropped and the repeated call plans anew
    u, _ = tp(th.cuda(), 0.0, vs, ep, **kw)
    assert tp.choleskyInfo["mode"] == "static"
    tp.check()
    tp.pivotHints = False            # (four HINTED block steps reach the 128 columns of this plan and a residual just inside the bound)
    tp._chol_static["rmax"], tp._chol_static["nsteps"] = 128, 4
    tp._chol_static["host"][0] = 50                      # (as if the previous call had fitted this plan ...
    tp._chol_static["head"].view(torch.int32)[7] = 1     #  ... in one block step: 1 + 3 to spare)
    u, _ = tp(th.cuda(), 0.0, vs, ep, **kw)
    assert tp.choleskyInfo["mode"] == "static" and tp.choleskyInfo["rmax"] == 128
    with pytest.raises(FloatingPointError, match="outgrew the static plan"):
        tp.check()
    assert tp._chol_static is None
    u, _ = tp(th.cuda(), 0.0, vs, ep, **kw)
    tp.check()
    assert tp.choleskyInfo["mode"] == "planning" and tp._chol_static["rmax"] >= 256 and float(tp.solverResidual) < 1e-12
    # the same failure inside an integrator: the stepper repeats the step once (same state, same step size) and hands back the same
    # result as with the full factorisation; and a synchronous call (observables on) repairs itself before it returns
    from vmc_pde_amd import FixedStepper
    ref_step = FixedStepper(timeStep=1e-9, mode="Heun", maxStep=1e-9, increase_fac=1.0)
    y_d, _, _ = ref_step.step(0.0, V.TDVP(diagonalShift=1e-4, svdTol=1e-6, solver="cholesky", choleskyEngine="dense"), th.cuda(), evolutionEq=ep, psi=vs, **{k: v for k, v in kw.items() if k != "timings"}, timings=None)
    tp._chol_static["rmax"], tp._chol_static["nsteps"] = 128, 4
    tp._chol_static["host"][0] = 50
    tp._chol_static["head"].view(torch.int32)[7] = 1
    st = FixedStepper(timeStep=1e-9, mode="Heun", maxStep=1e-9, increase_fac=1.0)
    y_l, dt_l, _ = st.step(0.0, tp, th.cuda(), evolutionEq=ep, psi=vs, **{k: v for k, v in kw.items() if k != "timings"}, timings=None)
    assert dt_l == 1e-9 and tp.choleskyInfo["mode"] in ("planning", "static") and tp._chol_static["rmax"] >= 256
    assert float(torch.linalg.norm((y_l - th.cuda()) - (y_d - th.cuda())) / torch.linalg.norm(y_d - th.cuda())) < 1e-4
    tp._chol_static["rmax"], tp._chol_static["nsteps"] = 128, 4
    tp._chol_static["host"][0] = 50
    tp._chol_static["head"].view(torch.int32)[7] = 1
    u_sync, info_sync = tp(th.cuda(), 0.0, vs, ep, nSamplesTDVP=N, nSamplesObs=N, timings=None, integrals=False)     # observables on: synchronous
    assert tp.choleskyInfo["mode"] == "planning" and float(tp.solverResidual) < 1e-12 and "entropy" in info_sync
    du = u_sync - u_d
    assert float(torch.linalg.norm(S @ du + D * du)) < 1e-11 * nF
    # a floor far below what the factorisation leaves of S: refinement would not contract, the full matrix is factored instead
    tq = V.TDVP(diagonalShift=1e-6, svdTol=1e-12, solver="cholesky")
    uq, _ = tq(th.cuda(), 0.0, vs, ep, **kw)
    assert tq.choleskyInfo["engine"] == "dense" and tq.choleskyInfo["tried"] == "lowrank" and tq._chol_backoff > 0
    assert tq.choleskyInfo["trace_left"] / tq.choleskyInfo["min_shift"] > 1e-4
    tq(th.cuda(), 0.0, vs, ep, **kw)
    assert tq.choleskyInfo == dict(engine="dense")                      # not retried right away
    with pytest.raises(ValueError, match="choleskyEngine"):
        V.TDVP(choleskyEngine="qr")
    monkeypatch.setattr(T, "LOWRANK_CHOL_MIN_P", 8192)                     # below the size threshold "auto" factors the full matrix
    ts = V.TDVP(diagonalShift=1e-4, svdTol=1e-6, solver="cholesky")
    us, _ = ts(th.cuda(), 0.0, vs, ep, **kw)
    assert ts.choleskyInfo == dict(engine="dense") and torch.equal(us, u_d)


def test_low_precision_gram_variants_keep_the_dense_solve(headline_net):
    """An S built from four int8 digits of an fp32 Jacobian (or from fp32 block sums) is 1e-8 away from the exact Gram: below that it has
    no numerical rank to reveal, and the low-rank routes must not be taken (a plan made on the first, fp64-built S of a run once
    failed the residual guard on the first int8-built one).  The six-digit form (1e-13) keeps them with a tolerance of 1e-12."""
    import vmc_pde_amd as V
    from helpers import make_spec, random_theta
    varstate, ep, th = headline_net
    N = 8192
    kw = dict(nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
    spec = make_spec(2, 4, (21, 21), latent="Gauss", coupling="no_add", offset=[1.0, 1.0], seed=4)
    g = torch.Generator().manual_seed(9)
    z = O.latent_from_normals(spec, th, torch.randn(N, 2, generator=g, dtype=torch.float64), None)
    vs32 = make_varstate(spec, th, sampler=FixedSampler(z.cuda()), jacobian_dtype=torch.float32)
    tp = XTDVP(diagonalShift=1e-4, svdTol=1e-6, solver="cholesky", int8Gram=True)
    seen = []
    for _ in range(5):
        tp(th.cuda(), 0.0, vs32, ep, **kw)
        tp.check()
        seen.append((tp._i8_digits, tp._lowrank_tol(), tp.choleskyInfo["engine"]))
        assert float(tp.solverResidual) < 1e-10
    assert seen[-1] == (4, None, "dense") and seen[-2][2] == "dense", seen
    vs64 = varstate(N)
    t6 = XTDVP(diagonalShift=1e-4, svdTol=1e-6, solver="cholesky", int8Gram=True)
    for _ in range(5):
        t6(th.cuda(), 0.0, vs64, ep, **kw)
        t6.check()
    assert t6._i8_digits == 6 and t6._lowrank_tol() == 1e-12 and t6.choleskyInfo["engine"] == "lowrank" and float(t6.solverResidual) < 1e-11
    tb = V.TDVP(diagonalShift=1e-3, svdTol=1e-6, solver="cholesky", f32BlockGram=True)
    for _ in range(3):
        tb(th.cuda(), 0.0, vs32, ep, **kw)
        tb.check()
    assert tb._lowrank_tol() is None and tb.choleskyInfo == dict(engine="dense")


def test_matrix_free_step_matches_the_assembled_one(headline_net):
    """EXPERIMENT `TDVP(matrixFree=True)`: the regularised solve with S = <J^T J> never assembled (its diagonal, the columns the pivoted
    factorisation asks for and products S0 v come from passes over the Jacobian) against the assembled path on the same samples."""
    import vmc_pde_amd as V
    varstate, ep, th = headline_net
    N = 8192
    vs = varstate(N)
    kw = dict(nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
    td = V.TDVP(diagonalShift=1e-4, svdTol=1e-6, solver="cholesky", choleskyEngine="dense")
    u_d, _ = td(th.cuda(), 0.0, vs, ep, **kw)
    td.check()
    S, F = td.S0.clone(), td.F0.clone()
    D = 1e-4 * torch.diagonal(S) + 1e-6 * torch.diagonal(S).max()
    nF = float(torch.linalg.norm(F))
    tm = XTDVP(diagonalShift=1e-4, svdTol=1e-6, solver="cholesky", matrixFree=True)
    for call in range(3):
        u, _ = tm(th.cuda(), 0.0, vs, ep, **kw)
        tm.check()
        assert tm.choleskyInfo["engine"] == "matrix-free" and tm.choleskyInfo["mode"] == ("planning" if call == 0 else "static"), tm.choleskyInfo
        assert tm._S0 is None and tm._S0_lazy                           # nothing assembled by the call ...
        if call == 2:                                                   # ... the attribute forms it on first access, from the retained Jacobian
            assert float((tm.S0 - S).abs().max()) < 1e-12 * float(S.abs().max()) and tm._S0 is not None
        assert float(torch.linalg.norm(tm.F0 - F)) < 1e-12 * nF
        du = u - u_d
        assert float(torch.linalg.norm(S @ du + D * du)) < 1e-10 * nF, float(torch.linalg.norm(S @ du + D * du)) / nF
        assert abs(float(tm.solverResidual)) < 1e-11 and abs(float(tm.tdvp_error) - float(td.tdvp_error)) < 1e-10
        assert abs(float(tm.ElocMean) - float(td.ElocMean)) < 1e-12 * (1 + abs(float(td.ElocMean)))
        assert abs(float(tm.ElocVar) - float(td.ElocVar)) < 1e-11 * (1 + abs(float(td.ElocVar)))
        # from the second call on the columns of S at the previous call's pivots are formed AHEAD in one pass over the Jacobian, and the
        # block steps that take their candidates from that list need no pass of their own (counted by the kernel, read one call later)
        assert tm.choleskyInfo["columns_ahead"] == (0 if call == 0 else -(-tm._mf_plan.get("prev_rank", tm._mf_plan["rank"]) // 32) * 32)
        if call == 2:
            assert tm.choleskyInfo["own_passes_last_call"] <= 2, tm.choleskyInfo
    # ... without them every block step is a pass of its own (and the result the same)
    tn = XTDVP(diagonalShift=1e-4, svdTol=1e-6, solver="cholesky", matrixFree=True, pivotHints=False)
    for call in range(3):
        un, _ = tn(th.cuda(), 0.0, vs, ep, **kw)
        tn.check()
    du = un - u_d
    assert tn.choleskyInfo["columns_ahead"] == 0 and tn.choleskyInfo["own_passes_last_call"] >= 8, tn.choleskyInfo
    assert float(torch.linalg.norm(S @ du + D * du)) < 1e-10 * nF and abs(float(tn.solverResidual)) < 1e-11


def test_matrix_free_spectral_solve_matches_the_assembled_one(headline_net):
    """`TDVP(matrixFree=True)` with the reference's default solver: the spectral decomposition from the factor of the never-assembled S
    (Q1^T S Q1 from one more pass over the Jacobian) against the dense one of the assembled S on the same samples."""
    import vmc_pde_amd as V
    varstate, ep, th = headline_net
    N = 8192
    vs = varstate(N)
    kw = dict(nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
    td = V.TDVP(svdTol=1e-6, spectralEngine="dense")
    u_d, _ = td(th.cuda(), 0.0, vs, ep, **kw)
    td.check()
    S, F, ev_d = td.S0.clone(), td.F0.clone(), td.ev.clone()
    nF, lmax, P = float(torch.linalg.norm(F)), float(ev_d[-1]), S.shape[0]
    tm = XTDVP(svdTol=1e-6, matrixFree=True)
    for call in range(2):
        u, _ = tm(th.cuda(), 0.0, vs, ep, **kw)
        tm.check()
        assert tm.spectralInfo["engine"] == "matrix-free" and tm._S0 is None and 1 <= tm.spectralInfo["rank"] < P // 6
        assert float((tm.ev - ev_d).abs().max()) < 1e-12 * lmax
        Vm = tm.V
        assert float((Vm.T @ Vm - torch.eye(P, dtype=torch.float64, device="cuda")).abs().max()) < 1e-13
        assert float(torch.linalg.norm(S @ Vm - Vm * tm.ev)) < 1e-11 * lmax
        assert float(torch.linalg.norm(S @ (u - u_d))) < 1e-9 * nF
        assert abs(float(tm.solverResidual) - float(td.solverResidual)) < 1e-9 and abs(float(tm.tdvp_error) - float(td.tdvp_error)) < 1e-9
    with pytest.raises(NotImplementedError, match="diagonalShift"):
        XTDVP(svdTol=1e-6, diagonalShift=1e-3, matrixFree=True)(th.cuda(), 0.0, vs, ep, **kw)


@pytest.mark.parametrize("P,n,ldj", [(37, 1000, 1000), (300, 4097, 4160), (1031, 8192, 8192), (5, 3, 3)])
def test_jt_matvec_against_numpy(P, n, ldj):
    """vmcpde_jt_matvec: y = JT (JT^T v) and z = JT^T v, the S0 v of the matrix-free step (ragged sample counts, odd strides: the
    scalar-load path; even strides: 16-byte loads)."""
    from vmc_pde_amd import _native as nat
    rng = np.random.default_rng(P + n)
    JT = np.zeros((P, ldj))
    JT[:, :n] = rng.standard_normal((P, n))
    JT[:, n:] = np.nan                                  # (pad columns are never read)
    v = rng.standard_normal(P)
    for off in (0, 1):                                  # (a base address that is not 16-byte aligned takes the scalar path)
        buf = torch.zeros(P * ldj + 1, dtype=torch.float64, device="cuda")
        Jd = buf[off:off + P * ldj].view(P, ldj)
        Jd.copy_(torch.from_numpy(JT))
        z = torch.empty(n, dtype=torch.float64, device="cuda")
        y = torch.empty(P, dtype=torch.float64, device="cuda")
        need = nat.lib().vmcpde_jt_matvec_workspace_bytes(P, n)
        ws = torch.empty(need // 8, dtype=torch.float64, device="cuda")
        nat.call("vmcpde_jt_matvec", nat.ptr(Jd), ldj, P, n, nat.ptr(torch.from_numpy(v).cuda()), nat.ptr(z), nat.ptr(y), nat.ptr(ws), need, nat.stream())
        torch.cuda.synchronize()
        z_ref = JT[:, :n].T @ v
        y_ref = JT[:, :n] @ z_ref
        assert np.abs(z.cpu().numpy() - z_ref).max() < 1e-13 * np.abs(z_ref).max()
        assert np.abs(y.cpu().numpy() - y_ref).max() < 1e-13 * np.abs(y_ref).max()


@pytest.mark.parametrize("hidden", [(3,), (6, 6)])
def test_lowrank_routes_on_small_models(hidden):
    """A model with fewer parameters than the 128-column pad of the static plan (P = 37 / ~200): the forced low-rank solve and the
    matrix-free step over several calls (planning, static, hints, columns ahead) against the dense factorisation."""
    import vmc_pde_amd as V
    spec = make_spec(2, 2, hidden, latent="Gauss", coupling="no_add", offset=[1.0, 1.0], seed=3)
    th = random_theta(spec, 3, out_scale=0.05)
    ep = V.EvolutionEquation(dim=2, name="advection_hamiltonian_wDiss")
    N = 4096
    g = torch.Generator().manual_seed(5)
    z = O.latent_from_normals(spec, th, torch.randn(N, 2, generator=g, dtype=torch.float64), None)
    vs = make_varstate(spec, th, sampler=FixedSampler(z.cuda()))
    kw = dict(nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
    td = V.TDVP(diagonalShift=1e-4, svdTol=1e-6, solver="cholesky", choleskyEngine="dense")
    u_d, _ = td(th.cuda(), 0.0, vs, ep, **kw)
    td.check()
    S, F = td.S0.clone(), td.F0.clone()
    P = S.shape[0]
    assert P < 256
    D = 1e-4 * torch.diagonal(S) + 1e-6 * torch.diagonal(S).max()
    nF = float(torch.linalg.norm(F))
    for flags in (dict(choleskyEngine="lowrank"), dict(matrixFree=True)):
        tp = XTDVP(diagonalShift=1e-4, svdTol=1e-6, solver="cholesky", **flags)     # (the product class with both experiments off for the first)
        modes = []
        for call in range(4):
            u, _ = tp(th.cuda(), 0.0, vs, ep, **kw)
            tp.check()
            modes.append((tp.choleskyInfo["engine"], tp.choleskyInfo.get("mode")))
            if tp.choleskyInfo["engine"] == "dense":        # (full rank within the cap: the forced engine hands over, as at any size)
                break
            du = u - u_d
            assert float(torch.linalg.norm(S @ du + D * du)) < 1e-10 * nF, (flags, call, modes)
            assert abs(float(tp.solverResidual)) < 1e-11
        assert modes[0][0] in ("lowrank", "matrix-free", "dense"), modes
        if modes[0][0] == "matrix-free":
            assert [m for _, m in modes] == ["planning", "static", "static", "static"], modes
        elif modes[0][0] == "lowrank":      # (the assembled route keeps planning when its buffers -- P rows -- cannot hold the 128-column pad)
            assert all(m in ("planning", "static") for _, m in modes), modes
    # the reference's spectral solve, matrix-free, on the same small model
    ts = V.TDVP(svdTol=1e-6, spectralEngine="dense")
    us, _ = ts(th.cuda(), 0.0, vs, ep, **kw)
    ts.check()
    tm = XTDVP(svdTol=1e-6, matrixFree=True)
    for call in range(3):
        um, _ = tm(th.cuda(), 0.0, vs, ep, **kw)
        tm.check()
        assert tm.spectralInfo["engine"] == "matrix-free"
        assert float((tm.ev - ts.ev).abs().max()) < 1e-12 * float(ts.ev[-1])
        assert float(torch.linalg.norm(S @ (um - us))) < 1e-9 * nF



@pytest.mark.parametrize("r,triangular,decades", [(1, False, 7), (5, True, 7), (64, True, 7), (65, False, 7), (206, True, 7), (256, True, 7),
                                                  (300, True, 3), (600, False, 2)])
def test_jacobi_columns_against_numpy(r, triangular, decades):
    """vmcpde_jacobi_columns (csrc/jacobi.hip): one-sided Jacobi on the columns of an r x r matrix whose columns span seven decades (as the
    triangular factor R1 of the pivoted factor does): on return the columns are orthogonal to sqrt(rows) eps, their norms are the singular
    values (to the accuracy of each COLUMN's scale, not of the largest), and U diag(sigma^2) U^T is C C^T -- the r x r core of the
    rank-revealing spectral solve (tdvp.py:57-64 on range(L)) without a library eigensolver."""
    from vmc_pde_amd import _native as nat
    import ctypes as C
    rng = np.random.default_rng(r)
    A = rng.standard_normal((r, r))
    if triangular:
        A = np.triu(A)
    A = A * np.logspace(0, -decades, r)[None, :]                  # graded columns (beyond 256 rows a workgroup sweeps 32, 16 or 8 columns at a
                                                                  # time and strongly graded RANDOM columns need more than 30 sweeps: milder there)
    rows, cols = C.c_int32(), C.c_int32()
    nat.call("vmcpde_jacobi_shape", r, C.byref(rows), C.byref(cols))
    rows, cols = rows.value, cols.value
    assert rows % 64 == 0 and rows >= r and cols >= r and rows - r < 64
    CT = torch.zeros((cols, rows), dtype=torch.float64, device="cuda")
    CT[:r, :r] = torch.from_numpy(A.T.copy())
    counters = torch.full((40,), -1, dtype=torch.int32, device="cuda")
    nat.call("vmcpde_jacobi_columns", nat.ptr(CT), r, 30, nat.ptr(counters), nat.stream())
    cnt = counters.cpu().numpy()[:31]
    assert (cnt >= 0).all() and (cnt == 0).any(), cnt                # converged: a sweep that rotated nothing
    first_zero = int(np.argmax(cnt == 0))
    assert (cnt[first_zero:] == 0).all() and first_zero <= 29, cnt  # ... and nothing ran behind it (random graded columns need 10-17 sweeps, the model's factor 6-7)
    B = CT.cpu().numpy()[:r, :r].T                                   # columns u_j sigma_j
    assert not CT[r:].any() and not CT[:, r:].any()                  # the zero padding stays zero
    sig = np.linalg.norm(B, axis=0)
    order = np.argsort(sig)[::-1]
    sv = np.linalg.svd(A, compute_uv=False)
    assert np.abs(sig[order] - sv).max() < (1e-14 + 4e-16 * r) * sv[0]                  # (rotations of ~10 sweeps: a few eps sqrt(r) per sweep)
    big = sv > 1e-9 * sv[0]
    assert np.abs(sig[order][big] / sv[big] - 1).max() < 1e-10       # relative accuracy well below the top of the spectrum
    U = B / sig
    G = U.T @ U
    assert np.abs(G - np.eye(r)).max() < 4 * np.sqrt(rows) * 1.2e-16 + 1e-15
    assert np.linalg.norm((U * sig ** 2) @ U.T - A @ A.T) < (1e-13 + 5e-15 * r) * sv[0] ** 2
    with pytest.raises(nat.VmcpdeError, match="1024"):
        nat.call("vmcpde_jacobi_columns", nat.ptr(CT), 1025, 4, nat.ptr(counters), nat.stream())


def test_jacobi_core_matches_the_library_core_at_headline_size(headline_net):
    """The same rank-revealing decomposition with the r x r core solved by the own Jacobi kernel (default) and by the library's syevd:
    eigenvalues to 1e-13 lambda_max, orthonormal V, the same update as the matrix sees it, twice (the second call launches the sweep
    count the first one needed).  The library core is the default: the own kernel is slower."""
    import vmc_pde_amd as V
    varstate, ep, th = headline_net
    N = 8192
    vs = varstate(N)
    out = {}
    names = {"library": "syevd", "jacobi": "jacobi", "auto": "sytrd(own)+stedc+back-transform(own)"}
    for core in ("library", "jacobi", "auto"):
        tp = V.TDVP(svdTol=1e-6, coreEigensolver=core)
        for _ in range(2):
            u, _ = tp(th.cuda(), 0.0, vs, ep, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
            tp.check()
        assert tp.spectralInfo["engine"] == "lowrank" and tp.spectralInfo["core"] == names[core], tp.spectralInfo
        out[core] = (tp.S0.clone(), tp.F0.clone(), tp.ev.clone(), tp.V.clone(), u.clone(), float(tp.solverResidual), float(tp.tdvp_error), tp)
    S, F, ev_l, V_l, u_l, res_l, err_l, _ = out["library"]
    P, lmax, nF = S.shape[0], float(ev_l[-1]), float(torch.linalg.norm(F))
    eye = torch.eye(P, dtype=torch.float64, device="cuda")
    r_l = float(torch.linalg.norm(S @ V_l - V_l * ev_l)) / lmax
    for core in ("jacobi", "auto"):       # "auto" = the default: own tridiagonalisation + library stedc + own back-transformation (r <= 256)
        S2, F2, ev_j, V_j, u_j, res_j, err_j, tj = out[core]
        assert torch.equal(S, S2) and torch.equal(F, F2)
        assert float((ev_j - ev_l).abs().max()) < 1e-13 * lmax and bool((ev_j[1:] >= ev_j[:-1]).all()), core
        assert float((V_j.T @ V_j - eye).abs().max()) < 1e-13, core
        r_j = float(torch.linalg.norm(S @ V_j - V_j * ev_j)) / lmax
        assert r_j < max(3 * r_l, 1e-13), (core, r_j, r_l)
        assert float(torch.linalg.norm(S @ (u_j - u_l))) < 1e-10 * nF, core
        assert abs(res_j - res_l) < 1e-10 and abs(err_j - err_l) < 1e-10, core
    tt = out["auto"][-1]
    assert tt._tridiag["off"] is False and int(tt._tridiag["host"][0]) == 0            # stedc's status word of the last call
    tj = out["jacobi"][-1]
    st = tj._jacobi
    assert not st["off"] and 4 <= st["n_sweeps"] <= 14, st            # the plan follows what the first call needed


@pytest.mark.parametrize("use_snr", [False, True])
def test_rhovar_from_the_ritz_vectors_is_the_dense_formula(headline_net, monkeypatch, use_snr):
    """tdvp.py:68-71 after the rank-revealing decomposition: only the r Ritz vectors can carry variance (v^T S v = 0 means every sample's
    centred gradient is orthogonal to v), so rhoVar comes from one r x P x N product instead of the weighted Gram and a P^3 product -- the
    same numbers as the dense formula on the Ritz modes, exact zeros where the dense formula leaves round-off; snr likewise; and the
    SNR-regularised solve (useSNR) gives the same update with either."""
    import vmc_pde_amd as V
    varstate, ep, th = headline_net
    N = 8192
    vs = varstate(N)
    kw = dict(nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
    tp = V.TDVP(svdTol=1e-6, useSNR=use_snr)
    u_fast, _ = tp(th.cuda(), 0.0, vs, ep, **kw)
    assert tp.spectralInfo["engine"] == "lowrank" and tp._ritz is not None
    rho_fast, snr_fast = tp.rhoVar.clone(), tp.snr.clone()
    monkeypatch.setenv("VMCPDE_RHOVAR_LOWRANK", "0")
    tp._rhoVar = tp._snr = None
    rho_dense = tp.rhoVar.clone()
    td = V.TDVP(svdTol=1e-6, useSNR=use_snr)
    u_dense, _ = td(th.cuda(), 0.0, vs, ep, **kw)
    snr_dense = td.snr.clone()
    assert torch.equal(td.ev, tp.ev)
    P, r = tp.S0.shape[0], tp.spectralInfo["rank"]
    ritz = tp.ev != 0
    assert int(ritz.sum()) in (r, r - 1, r - 2)                       # (a Ritz value may be an exact 0 only by accident)
    top = float(rho_dense.abs().max())
    assert float((rho_fast - rho_dense)[ritz].abs().max()) < 1e-10 * top
    assert not rho_fast[~ritz].any() and float(rho_dense[~ritz].abs().max()) < 1e-10 * top        # round-off against exact zeros
    assert not snr_fast[~ritz].any() and bool(torch.isfinite(snr_fast).all())
    live = ritz & (rho_dense > 1e-7 * top)
    assert int(live.sum()) >= 30
    assert float(((snr_fast - snr_dense)[live] / snr_dense[live]).abs().max()) < 1e-5
    nF = float(torch.linalg.norm(tp.F0))
    assert float(torch.linalg.norm(tp.S0 @ (u_fast - u_dense))) < 1e-9 * nF         # (identical when useSNR is off: rhoVar is not in the solve)
    if not use_snr:
        assert torch.equal(u_fast, u_dense)


@pytest.mark.parametrize("n", [1, 2, 3, 17, 64, 65, 128, 129, 206, 255, 256, 257, 263, 320, 384])
def test_own_tridiagonalisation_and_back_transformation_against_numpy(n):
    """vmcpde_sytrd_small / vmcpde_apply_reflectors (csrc/tridiag.hip): T = Q tridiag(d, e) Q^T of a symmetric matrix graded over seven
    decades (like the core Q1^T S Q1 of the rank-revealing spectral solve) in one workgroup -- the reflectors give an orthogonal Q, Q^T T Q
    is tridiagonal with exactly the returned (d, e), the eigenvalues of the tridiagonal matrix are those of T; and the back-transformation of
    the tridiagonal matrix's eigenvectors (numpy's here) gives eigenvectors of T.  Bitwise reproducible (no atomics)."""
    from vmc_pde_amd import _native as nat
    rng = np.random.default_rng(n)
    Q0, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = 2.0e4 * 10.0 ** (-np.arange(n) / max(n / 7.0, 1.0))
    A = (Q0 * lam) @ Q0.T
    A = 0.5 * (A + A.T)
    ld = 264 if n <= 256 else 392                                     # a padded leading dimension (only the lower triangle is read)
    T = torch.full((n, ld), float("nan"), dtype=torch.float64, device="cuda")
    T[:, :n] = torch.from_numpy(np.tril(A) + np.triu(np.full((n, n), np.nan), 1))     # the strict upper triangle is never read
    assert int(nat.lib().vmcpde_sytrd_small_max_n()) == 256 and int(nat.lib().vmcpde_sytrd_medium_max_n()) == 384
    # n <= 256: the one-workgroup kernel; 257 .. 384 (ABI 7; BASELINE C3's core has rank 263): n - 256 leading steps of two launches each on a
    # symmetric copy in the workspace, then the same kernel on the trailing block -- through vmcpde_sytrd_medium, which IS the small entry below 257
    nmax = 256 if n <= 256 else 384
    nwork = max(int(nat.lib().vmcpde_sytrd_medium_workspace_bytes(n)) // 8, 1)
    assert (nwork == 1) == (n <= 256)
    outs = []
    for rep in range(2):
        D, E, tau = (torch.full((nmax,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(3))
        Vr = torch.full((nmax, nmax), float("nan"), dtype=torch.float64, device="cuda")
        work = torch.full((nwork,), float("nan"), dtype=torch.float64, device="cuda")
        if n <= 256 and rep == 0:
            nat.call("vmcpde_sytrd_small", nat.ptr(T), ld, n, nat.ptr(D), nat.ptr(E), nat.ptr(tau), nat.ptr(Vr), nmax, nat.stream())
        else:
            nat.call("vmcpde_sytrd_medium", nat.ptr(T), ld, n, nat.ptr(D), nat.ptr(E), nat.ptr(tau), nat.ptr(Vr), nmax, nat.ptr(work), nwork * 8, nat.stream())
        outs.append((D.cpu().numpy()[:n], E.cpu().numpy()[:n], tau.cpu().numpy()[:n], Vr.cpu().numpy()[:n, :n]))
    d, e, t, Vh = outs[0]
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)                                   # a pure function of the input bits
    assert np.isfinite(d).all() and np.isfinite(e).all() and np.isfinite(t).all() and np.isfinite(Vh).all()
    assert e[n - 1] == 0.0 and t[n - 1] == 0.0
    Q = np.eye(n)
    for k in range(n - 1):                                            # Q = H_0 H_1 ... H_{n-2}
        v = Vh[k]
        assert not v[:k + 1].any() and v[k + 1] == 1.0
        Q = Q - t[k] * np.outer(Q @ v, v)
    assert np.abs(Q.T @ Q - np.eye(n)).max() < 1e-13
    Tt = np.diag(d) + np.diag(e[:n - 1], -1) + np.diag(e[:n - 1], 1)
    scale = np.abs(A).max()
    assert np.abs(Q.T @ A @ Q - Tt).max() < 2e-13 * scale
    w_ref = np.linalg.eigvalsh(A)
    assert np.abs(np.linalg.eigvalsh(Tt) - w_ref).max() < 1e-13 * w_ref[-1]
    # back-transformation: rows of Zt = eigenvectors of the tridiagonal matrix
    wt, Zt = np.linalg.eigh(Tt)
    Zd = torch.from_numpy(np.ascontiguousarray(Zt.T)).cuda()
    Vd = torch.from_numpy(np.ascontiguousarray(np.pad(Vh, ((0, nmax - n), (0, nmax - n))))).cuda()      # (kept alive until the kernel has run)
    td = torch.from_numpy(np.pad(t, (0, nmax - n))).cuda()
    nat.call("vmcpde_apply_reflectors", nat.ptr(Vd), nmax, nat.ptr(td), n, nat.ptr(Zd), n, n, nat.stream())
    W = Zd.cpu().numpy().T                                            # columns = eigenvectors of A
    assert np.abs(W - Q @ Zt).max() < 1e-13
    assert np.linalg.norm(A @ W - W * wt) < 1e-12 * w_ref[-1] and np.abs(W.T @ W - np.eye(n)).max() < 1e-13
    with pytest.raises(nat.VmcpdeError, match="256"):
        nat.call("vmcpde_sytrd_small", nat.ptr(T), ld, 257, nat.ptr(D), nat.ptr(E), nat.ptr(tau), nat.ptr(Vr), nmax, nat.stream())
    with pytest.raises(nat.VmcpdeError, match="384"):
        nat.call("vmcpde_sytrd_medium", nat.ptr(T), ld, 385, nat.ptr(D), nat.ptr(E), nat.ptr(tau), nat.ptr(Vr), nmax, nat.ptr(work), nwork * 8, nat.stream())


@pytest.mark.parametrize("n", [3, 65, 206, 256, 263, 384])
def test_tridiag_core_route_against_the_library_eigh(n):
    """The whole core route of `coreEigensolver="auto"` on a graded matrix: own tridiagonalisation, rocSOLVER's stedc through ctypes
    (vmc_pde_amd/rocsolver.py), own back-transformation -- against torch.linalg.eigh."""
    import vmc_pde_amd as V
    g = torch.Generator().manual_seed(n)
    Q0, _ = torch.linalg.qr(torch.randn(n, n, generator=g, dtype=torch.float64))
    lam = 2.0e4 * 10.0 ** (-torch.arange(n, dtype=torch.float64) / max(n / 7.0, 1.0))
    A = ((Q0 * lam) @ Q0.T).cuda()
    A = 0.5 * (A + A.T)
    tp = V.TDVP()
    assert tp._core_choice(n) == "tridiag"
    w = {}
    wv, W = tp._core_eigh_tridiag(w, A, n, A.device)
    W = W.clone()                                    # (W is a view of the reused workspace: keep the first call's values, not an alias)
    wv2, W2 = tp._core_eigh_tridiag(w, A, n, A.device)
    assert W.data_ptr() != W2.data_ptr()
    assert torch.equal(wv, wv2) and torch.equal(W, W2)
    ev, _ = torch.linalg.eigh(A)
    lmax = float(ev[-1])
    assert float((wv - ev).abs().max()) < 1e-13 * lmax and bool((wv[1:] >= wv[:-1]).all())
    eye = torch.eye(n, dtype=torch.float64, device="cuda")
    assert float((W.T @ W - eye).abs().max()) < 1e-13
    assert float(torch.linalg.norm(A @ W - W * wv)) < 1e-12 * lmax
    assert tp._core_choice(n) == "tridiag" and int(tp._tridiag["host"][0]) == 0      # (reads the status word of the call before)
    assert tp._core_choice(257) == "tridiag" and tp._core_choice(385) == "library" and tp._core_choice(2) == "library"
    assert V.TDVP(coreEigensolver="library")._core_choice(n) == "library"


def test_a_failed_stedc_rejects_the_step_it_belongs_to(monkeypatch):
    """The status word of the core's tridiagonal eigensolver is part of the step's health record: a non-zero word makes `check()` raise for THAT
    step (as torch.linalg.eigh raises on a non-zero info) and sends later calls to the library core -- it is not merely a warning one call later."""
    import vmc_pde_amd as V
    from vmc_pde_amd import rocsolver
    from helpers import make_spec, random_theta, make_varstate, FixedSampler
    if not rocsolver.available():
        pytest.skip("rocSOLVER's stedc is not resolvable")
    dim, N = 2, 4096
    spec = make_spec(dim, 4, (3, 3), offset=[0.5, -0.5])           # (P = 181: any rank fits the own core)
    th = random_theta(spec, 3)
    z = torch.randn(N, dim, generator=torch.Generator().manual_seed(1), dtype=torch.float64).cuda()
    vs = make_varstate(spec, th, sampler=FixedSampler(z))
    eq = V.EvolutionEquation(dim=dim, name="advection_hamiltonian_wDiss")
    tp = V.TDVP(solver="eigh", spectralEngine="lowrank", svdTol=1e-6)
    upd, _ = tp(th.cuda(), 0.0, vs, eq, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
    tp.check()                                                         # a healthy call passes, on the own core
    assert tp.spectralInfo["core"].startswith("sytrd(own)")
    real = rocsolver.stedc

    def failing(D, E, Z, info):
        real(D, E, Z, info)
        info.fill_(3)
    monkeypatch.setattr(rocsolver, "stedc", failing)
    upd, _ = tp(th.cuda(), 0.0, vs, eq, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
    with pytest.raises(FloatingPointError, match="stedc"):
        tp.check()
    monkeypatch.setattr(rocsolver, "stedc", real)
    upd2, _ = tp(th.cuda(), 0.0, vs, eq, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
    tp.check()
    assert tp.spectralInfo["core"] == "syevd"                          # the core stays on the library solver afterwards
    assert float((upd2 - upd).abs().max()) < 1e-8 * float(upd.abs().max())


@pytest.mark.parametrize("P,r,ldl", [(4213, 256, 4224), (300, 7, 320), (65, 64, 128), (16181, 384, 16192)])
def test_woodbury_application_against_numpy(P, r, ldl):
    """vmcpde_woodbury_project / _combine: M^-1 v for M = D + L L^T through G = I + L^T D^-1 L (factored and solved by the library's own small
    Cholesky), for v = F and for v = the residual F - S u - D u of a refinement step -- against numpy's dense solve of M."""
    from vmc_pde_amd import _native as nat
    rng = np.random.default_rng(P + r)
    L = rng.standard_normal((P, r)) * np.logspace(0, -3, r)[None, :]
    L[:, r - 2:] = 0.0                                                 # (columns past the rank are zero in the static plan)
    D = rng.uniform(1e-3, 1.0, P)
    F, u0, Su = rng.standard_normal(P), rng.standard_normal(P), rng.standard_normal(P)
    LT = torch.zeros((r, ldl), dtype=torch.float64, device="cuda")
    LT[:, :P] = torch.from_numpy(L.T.copy())
    Dd, Fd, ud, Sud = (torch.from_numpy(v).cuda() for v in (D, F, u0, Su))
    rpad = (r + 127) // 128 * 128
    G = torch.zeros((rpad, rpad), dtype=torch.float64, device="cuda")
    G[:r, :r] = torch.from_numpy(np.eye(r) + L.T @ (L / D[:, None]))
    G[r:, r:] = torch.eye(rpad - r, dtype=torch.float64, device="cuda")
    need = nat.lib().vmcpde_chol_workspace_bytes(rpad)
    gws = torch.empty(need // 8, dtype=torch.float64, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    nat.call("vmcpde_cholesky", nat.ptr(G), rpad, rpad, nat.ptr(info), nat.ptr(gws), need, nat.stream())
    y = torch.empty(P, dtype=torch.float64, device="cuda")
    t1 = torch.zeros(rpad, dtype=torch.float64, device="cuda")
    t2 = torch.empty(rpad, dtype=torch.float64, device="cuda")
    out = torch.empty(P, dtype=torch.float64, device="cuda")
    M = np.diag(D) + L @ L.T
    for u_in, v in ((None, F), (ud, F - Su - D * u0)):
        nat.call("vmcpde_woodbury_project", nat.ptr(LT), ldl, P, r, nat.ptr(Dd), nat.ptr(Fd), nat.ptr(Sud) if u_in is not None else None, nat.ptr(u_in),
                 nat.ptr(y), nat.ptr(t1), nat.stream())
        assert np.allclose(y.cpu().numpy(), v / D, rtol=1e-12, atol=1e-13)                # (the residual is formed with a fused multiply-add)
        assert np.allclose(t1[:r].cpu().numpy(), L.T @ (v / D), rtol=1e-11, atol=1e-12)
        nat.call("vmcpde_chol_solve", nat.ptr(G), rpad, rpad, nat.ptr(t1), nat.ptr(t2), nat.ptr(gws), need, nat.stream())
        target = out if u_in is None else ud.clone()
        nat.call("vmcpde_woodbury_combine", nat.ptr(LT), ldl, P, r, nat.ptr(Dd), nat.ptr(y), nat.ptr(t2), nat.ptr(target) if u_in is not None else None,
                 nat.ptr(target), nat.stream())                        # (in place over u for the refinement form)
        ref = np.linalg.solve(M, v) + (u0 if u_in is not None else 0.0)
        assert np.linalg.norm(target.cpu().numpy() - ref) < 1e-9 * np.linalg.norm(ref)
    with pytest.raises(nat.VmcpdeError, match="come together"):
        nat.call("vmcpde_woodbury_project", nat.ptr(LT), ldl, P, r, nat.ptr(Dd), nat.ptr(Fd), None, nat.ptr(ud), nat.ptr(y), nat.ptr(t1), nat.stream())


@pytest.mark.parametrize("P,r,rank", [(4213, 256, 207), (300, 64, 64), (1000, 128, 5), (16181, 384, 300)])
def test_woodbury_prepare_and_gram_against_numpy(P, r, rank):
    """ABI 7: vmcpde_woodbury_prepare (D, 1 / D with a zero pad, the rows of LT behind the device-side rank zeroed) and vmcpde_woodbury_gram
    (G = I + L^T D^-1 L from 64 x 64 blocks x chunks of 256 parameters on the matrix pipe, deterministic) against numpy; and
    vmcpde_gather_words, the one record a static call hands to the host."""
    from vmc_pde_amd import _native as nat
    rng = np.random.default_rng(P + r)
    ldl = (P + 63) // 64 * 64
    LTh = rng.standard_normal((r + 3, ldl)) * np.logspace(0, -4, r + 3)[:, None]       # (stale columns everywhere, pad columns included)
    diagS = rng.uniform(0.1, 2.0, P)
    shift, floor = 1e-3, 2e-5
    LT = torch.from_numpy(LTh.copy()).cuda()
    state = torch.tensor([rank, 1], dtype=torch.int32, device="cuda")
    D = torch.full((P,), float("nan"), dtype=torch.float64, device="cuda")
    invD = torch.full((ldl,), float("nan"), dtype=torch.float64, device="cuda")
    fl = torch.tensor([floor], dtype=torch.float64, device="cuda")
    nat.call("vmcpde_woodbury_prepare", nat.ptr(torch.from_numpy(diagS).cuda()), shift, nat.ptr(fl), P, ldl, nat.ptr(D), nat.ptr(invD), nat.ptr(LT), nat.ptr(state), r,
             nat.stream())
    Dref = shift * diagS + floor
    assert np.allclose(D.cpu().numpy(), Dref, rtol=1e-15) and np.allclose(invD[:P].cpu().numpy(), 1.0 / Dref, rtol=1e-15)
    assert not invD[P:].any()
    L_after = LT.cpu().numpy()
    assert np.array_equal(L_after[:rank], LTh[:rank]) and not L_after[rank:r].any() and np.array_equal(L_after[r:], LTh[r:])
    need = nat.lib().vmcpde_woodbury_gram_workspace_bytes(ldl, r)
    ws = torch.empty(need // 8, dtype=torch.float64, device="cuda")
    outs = []
    for _ in range(2):
        G = torch.full((r, r), float("nan"), dtype=torch.float64, device="cuda")
        nat.call("vmcpde_woodbury_gram", nat.ptr(LT), ldl, r, nat.ptr(invD), nat.ptr(G), r, nat.ptr(ws), need, nat.stream())
        outs.append(G.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])                                  # fixed summation order
    Lr = L_after[:r, :P]
    Gref = np.eye(r) + (Lr / Dref) @ Lr.T
    assert np.abs(outs[0] - Gref).max() < 1e-12 * np.abs(Gref).max() and np.array_equal(outs[0], outs[0].T)
    with pytest.raises(nat.VmcpdeError, match="multiple of 64"):
        nat.call("vmcpde_woodbury_gram", nat.ptr(LT), ldl, 65, nat.ptr(invD), nat.ptr(G), r, nat.ptr(ws), need, nat.stream())
    # gather: four sources with a skipped one keep their slots
    a = torch.arange(4, dtype=torch.float64, device="cuda") + 0.5
    c = torch.tensor([7, -3], dtype=torch.int32, device="cuda")
    d = torch.arange(4, dtype=torch.float64, device="cuda") * 3.0
    dst = torch.full((16,), -1.0, dtype=torch.float64, device="cuda")
    srcs, counts = (nat.C.c_void_p * 8)(), (nat.C.c_int32 * 8)()
    for i, (src, n) in enumerate([(a, 4), (None, 1), (c, 1), (d, 4)]):
        srcs[i] = None if src is None else src.data_ptr()
        counts[i] = n
    nat.call("vmcpde_gather_words", srcs, counts, 4, nat.ptr(dst), nat.stream())
    h = dst.cpu()
    assert torch.equal(h[:4], a.cpu()) and float(h[4]) == -1.0 and h[5:6].view(torch.int32).tolist() == [7, -3] and torch.equal(h[6:10], d.cpu())


@pytest.mark.parametrize("P,q,rmax", [(600, 40, 128), (4213, 200, 256), (1500, 300, 384)])
def test_factor_on_a_frozen_pivot_set_against_numpy(P, q, rmax):
    """ABI 7: vmcpde_frozen_gather / vmcpde_cholesky_tall / vmcpde_frozen_finish / vmcpde_frozen_check -- the partial Cholesky of S on a GIVEN
    pivot set in a given order, S = L L^T + R with L = S[:, Pi] C^-T, C C^T = S[Pi, Pi]: against numpy's Cholesky of the pivot block and a
    triangular solve, the remainder's diagonal against diag(S - L L^T), and the bookkeeping (state, trace) a pivoted run would leave."""
    from vmc_pde_amd import _native as nat
    rng = np.random.default_rng(P + q)
    A = rng.standard_normal((P, q)) * np.logspace(0, -5, q)[None, :]
    S = A @ A.T + 1e-13 * np.eye(P)
    # a pivot set: greedy diagonal pivoting on S (numpy), r = q - 3 pivots, then the order slightly shuffled (a stale order)
    d, Lg, piv = np.diag(S).copy(), np.zeros((P, q)), []
    for k in range(q - 3):
        p = int(np.argmax(d))
        piv.append(p)
        col = (S[:, p] - Lg[:, :k] @ Lg[p, :k]) / np.sqrt(d[p])
        Lg[:, k] = col
        d -= col * col
        d[p] = -np.inf
    r = len(piv)
    for k in range(5, r - 1, 7):
        piv[k], piv[k + 1] = piv[k + 1], piv[k]
    piv = np.asarray(piv, dtype=np.int32)
    lds, ldl = P + 11, (P + 63) // 64 * 64
    Sd = torch.zeros((P, lds), dtype=torch.float64, device="cuda")
    Sd[:, :P] = torch.from_numpy(S)
    pivd = torch.zeros(rmax, dtype=torch.int32, device="cuda")
    pivd[:r] = torch.from_numpy(piv)
    rows = (rmax + P + 127) // 128 * 128
    T = torch.zeros((rows, rmax), dtype=torch.float64, device="cuda")
    T[:rmax + P].fill_(float("nan"))                       # every entry of the live rows is written by the gather
    nat.call("vmcpde_frozen_gather", nat.ptr(Sd), lds, P, nat.ptr(pivd), r, rmax, nat.ptr(T), rmax, nat.stream())
    Th = T.cpu().numpy()
    Cref = np.eye(rmax)
    Cref[:r, :r] = S[np.ix_(piv, piv)]
    Bref = np.zeros((P, rmax))
    Bref[:, :r] = S[:, piv]
    assert np.array_equal(Th[:rmax], Cref) and np.array_equal(Th[rmax:rmax + P], Bref) and not Th[rmax + P:].any()
    need = nat.lib().vmcpde_chol_workspace_bytes(rmax)
    ws = torch.empty(need // 8, dtype=torch.float64, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    nat.call("vmcpde_cholesky_tall", nat.ptr(T), rmax, rmax + P, rmax, nat.ptr(info), nat.ptr(ws), need, nat.stream())
    assert int(info.item()) == 0
    LT = torch.full((rmax, ldl), float("nan"), dtype=torch.float64, device="cuda")
    left = torch.full((P,), float("nan"), dtype=torch.float64, device="cuda")
    nat.call("vmcpde_frozen_finish", nat.ptr(T), rmax, nat.ptr(Sd), lds, P, rmax, nat.ptr(LT), ldl, nat.ptr(left), nat.stream())
    import scipy.linalg as sla
    Cc = np.linalg.cholesky(S[np.ix_(piv, piv)])
    Lref = sla.solve_triangular(Cc, S[:, piv].T, lower=True).T                 # P x r
    Lh = LT.cpu().numpy()
    assert not Lh[:, P:].any() and not Lh[r:].any()
    scale = np.sqrt(np.abs(S).max())
    assert np.abs(Lh[:r, :P].T - Lref).max() < 1e-9 * scale                    # (late columns: pivots of 1e-10 |S| amplify round-off)
    Rdiag = np.diag(S) - (Lref * Lref).sum(1)
    assert np.abs(left.cpu().numpy() - Rdiag).max() < 1e-12 * np.abs(S).max()
    assert np.abs(Lh[:r, :P].T @ Lh[:r, :P] - Lref @ Lref.T).max() < 1e-12 * np.abs(S).max()
    # check: the record of the call
    D = torch.from_numpy(rng.uniform(1e-4, 1.0, P)).cuda()
    state = torch.zeros(2, dtype=torch.int32, device="cuda")
    head = torch.zeros(8, dtype=torch.float64, device="cuda")
    out = torch.zeros(3, dtype=torch.float64, device="cuda")
    nat.call("vmcpde_frozen_check", nat.ptr(left), P, nat.ptr(D), r, nat.ptr(info), nat.ptr(state), nat.ptr(head), nat.ptr(out), nat.stream())
    lh = left.cpu().numpy()
    tr = lh[lh > 0].sum()
    assert state.tolist() == [r, 1] and float(head[1]) == pytest.approx(tr, rel=1e-12)
    o = out.cpu().numpy()
    assert o[0] == pytest.approx(tr / float(D.min()), rel=1e-12) and o[1] == 0.0 and o[2] == pytest.approx(lh.max(), rel=1e-14)
