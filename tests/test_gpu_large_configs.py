"""GPU tests at BASELINE config 3 (P = 16 181, hidden (43,43)) and config 4's per-GPU shard (d = 6, hidden (19,19), P = 4155,
32 768 samples): the code paths only large problems reach -- `chol_strip_kernel<32>` (P >~ 4350), the two-level (super-panel) schedule
of the blocked Cholesky (P > 7040), the one-workgroup-per-CU flow geometry of the (43,43) net, the Gram kernel without
sample splits.  Reference shapes: var_state.py:110-124; tdvp.py:36-94.

Oracle comparisons run at sizes the CPU oracle finishes in seconds; the full-size runs are checked through size-independent
properties (||L L^T v - A v||, S v = J^T (J v), additivity over sample splits, solver residual, pinned-pivot count)."""
import os

import numpy as np
import pytest
import torch

from vmc_pde_amd import _native as nat

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _pad(v, m):
    return (v + m - 1) // m * m


def _spd_gpu(P, seed, K=1024, ridge=0.05):
    """A = X X^T / K + ridge * diag(1..2): positive definite, condition ~1e2-1e3, built on the device (2 GiB at P = 16181)."""
    g = torch.Generator(device=DEV).manual_seed(seed)
    X = torch.randn((P, K), generator=g, dtype=torch.float64, device=DEV)
    A = X @ X.T / K
    A += torch.diag(ridge * (1.0 + torch.arange(P, dtype=torch.float64, device=DEV) / P))
    return A


def _factor(A, P):
    Pp = _pad(P, 128)
    Ap = torch.zeros((Pp, Pp), dtype=torch.float64, device=DEV)
    Ap[:P, :P] = A
    need = nat.lib().vmcpde_chol_workspace_bytes(P)
    ws = torch.empty(need // 8, dtype=torch.float64, device=DEV)
    info = torch.zeros(1, dtype=torch.int32, device=DEV)
    nat.call("vmcpde_cholesky", nat.ptr(Ap), Pp, P, nat.ptr(info), nat.ptr(ws), need, nat.stream())
    return Ap, Pp, ws, need, info


@pytest.mark.parametrize("P", [4400, 6300, 16181])
def test_cholesky_large_P_against_rocsolver(P):
    # 4400: chol_strip_kernel<32>; 6300: updates of more than 1024 tiles; 16181: BASELINE config 3 (super-panels of 8)
    A = _spd_gpu(P, P)
    Ap, Pp, ws, need, info = _factor(A, P)
    assert int(info.item()) == 0
    L = torch.tril(Ap[:P, :P])
    Lref = torch.linalg.cholesky(A)
    assert float((L - Lref).abs().max() / Lref.abs().max()) < 1e-11
    del Lref
    g = torch.Generator(device=DEV).manual_seed(1)
    v = torch.randn(P, generator=g, dtype=torch.float64, device=DEV)
    Av = A @ v
    assert float(torch.linalg.norm(L @ (L.T @ v) - Av) / torch.linalg.norm(Av)) < 1e-13
    # and the triangular solves at this size
    u = torch.empty(P, dtype=torch.float64, device=DEV)
    nat.call("vmcpde_chol_solve", nat.ptr(Ap), Pp, P, nat.ptr(Av), nat.ptr(u), nat.ptr(ws), need, nat.stream())
    assert float(torch.linalg.norm(u - v) / torch.linalg.norm(v)) < 1e-9


@pytest.mark.parametrize("P", [300, 1000])
def test_cholesky_side_stream_branch_on_small_matrices(P, monkeypatch):
    # the round-1 schedule (32-wide diagonal block, VMCPDE_CHOL_DIAG16=0) is kept for A/B: VMCPDE_CHOL_FUSED_MAX=1 sends every
    # trailing update with more than one tile through its fork/join branch (chol_syrk_kernel on the caller's stream, the next
    # diagonal block on the device's side stream)
    monkeypatch.setenv("VMCPDE_CHOL_DIAG16", "0")
    monkeypatch.setenv("VMCPDE_CHOL_FUSED_MAX", "1")
    rng = np.random.default_rng(P)
    Q, _ = np.linalg.qr(rng.standard_normal((P, P)))
    Amat = (Q * np.logspace(0, -4, P)) @ Q.T
    Ap, Pp, ws, need, info = _factor(torch.as_tensor(Amat).to(DEV), P)
    assert int(info.item()) == 0
    L = np.tril(Ap.cpu().numpy()[:P, :P])
    Lref = np.linalg.cholesky(Amat)
    assert np.abs(L - Lref).max() < 1e-10 * np.abs(Lref).max()
    monkeypatch.delenv("VMCPDE_CHOL_FUSED_MAX")
    Ap2, *_ = _factor(torch.as_tensor(Amat).to(DEV), P)
    assert torch.equal(Ap2[:P, :P].tril(), Ap[:P, :P].tril())       # fused and forked launches do the same arithmetic


@pytest.mark.parametrize("P,W,head", [(1000, 2, 4), (1000, 3, 5), (1500, 4, 6), (2049, 2, 1), (2049, 2, 40)])
def test_cholesky_super_panels_over_the_head_of_the_matrix_only(P, W, head, monkeypatch):
    # VMCPDE_CHOL_SUPER_HEAD: super-panels of W block columns over the first `head` block columns (the last one clipped at the head's
    # end), single panels behind -- same factor as the one-level schedule
    rng = np.random.default_rng(P + W + head)
    Q, _ = np.linalg.qr(rng.standard_normal((P, P)))
    Amat = (Q * np.logspace(0, -4, P)) @ Q.T
    monkeypatch.setenv("VMCPDE_CHOL_SUPER", str(W))
    monkeypatch.setenv("VMCPDE_CHOL_SUPER_HEAD", str(head))
    Ap, Pp, ws, need, info = _factor(torch.as_tensor(Amat).to(DEV), P)
    assert int(info.item()) == 0
    Lref = np.linalg.cholesky(Amat)
    assert np.abs(np.tril(Ap.cpu().numpy()[:P, :P]) - Lref).max() < 1e-10 * np.abs(Lref).max()
    monkeypatch.setenv("VMCPDE_CHOL_SUPER", "1")
    monkeypatch.delenv("VMCPDE_CHOL_SUPER_HEAD")
    Ap1, *_ = _factor(torch.as_tensor(Amat).to(DEV), P)
    assert float((Ap1[:P, :P].tril() - Ap[:P, :P].tril()).abs().max()) < 1e-12 * np.abs(Lref).max()


@pytest.mark.parametrize("P,W", [(300, 2), (1000, 2), (1000, 3), (1000, 8), (1500, 4), (2049, 16)])
def test_cholesky_super_panels_on_small_matrices(P, W, monkeypatch):
    # the two-level schedule large matrices use (super-panels of W block columns: K = 128 W trailing updates, chunked strips,
    # two tiles per workgroup), forced onto small ones; ragged last block, last super-panel shorter than W, W > number of panels
    rng = np.random.default_rng(P + W)
    Q, _ = np.linalg.qr(rng.standard_normal((P, P)))
    Amat = (Q * np.logspace(0, -4, P)) @ Q.T
    monkeypatch.setenv("VMCPDE_CHOL_SUPER", str(W))
    Ap, Pp, ws, need, info = _factor(torch.as_tensor(Amat).to(DEV), P)
    assert int(info.item()) == 0
    L = np.tril(Ap.cpu().numpy()[:P, :P])
    Lref = np.linalg.cholesky(Amat)
    assert np.abs(L - Lref).max() < 1e-10 * np.abs(Lref).max()
    v = rng.standard_normal(P)
    u = torch.empty(P, dtype=torch.float64, device=DEV)
    b = torch.as_tensor(Amat @ v).to(DEV)
    nat.call("vmcpde_chol_solve", nat.ptr(Ap), Pp, P, nat.ptr(b), nat.ptr(u), nat.ptr(ws), need, nat.stream())
    assert np.linalg.norm(u.cpu().numpy() - v) / np.linalg.norm(v) < 1e-7
    monkeypatch.setenv("VMCPDE_CHOL_SUPER", "1")
    Ap1, *_ = _factor(torch.as_tensor(Amat).to(DEV), P)
    assert float((Ap1[:P, :P].tril() - Ap[:P, :P].tril()).abs().max()) < 1e-12 * np.abs(Lref).max()


@pytest.mark.parametrize("P", [300, 1000, 4213])
def test_cholesky_one_launch_per_panel_schedule(P, monkeypatch):
    # VMCPDE_CHOL_PANEL=1: strips, next diagonal block and trailing tiles of a panel as counter-chained workgroups of one launch
    # (chol_panel_kernel; opt-in because it measured slower) -- same arithmetic as the default schedule up to the order in which
    # the first block column is updated
    A = _spd_gpu(P, P + 1)
    Ap0, *_ = _factor(A, P)
    monkeypatch.setenv("VMCPDE_CHOL_PANEL", "1")
    Ap1, Pp, ws, need, info = _factor(A, P)
    assert int(info.item()) == 0                      # (a timed-out wait would set bit 30)
    L0, L1 = torch.tril(Ap0[:P, :P]), torch.tril(Ap1[:P, :P])
    assert float((L0 - L1).abs().max() / L0.abs().max()) < 1e-13
    Lref = torch.linalg.cholesky(A)
    assert float((L1 - Lref).abs().max() / Lref.abs().max()) < 1e-11


def _gram_full(JT, ldj, P, N):
    npk = nat.lib().vmcpde_gram_packed_doubles(P)
    packed = torch.empty(npk, dtype=torch.float64, device=DEV)
    need = nat.lib().vmcpde_gram_workspace_bytes(P, N)
    ws = torch.empty(max(need // 8, 1), dtype=torch.float64, device=DEV)
    nat.call("vmcpde_gram", nat.ptr(JT), ldj, 0, P, N, None, None, nat.ptr(packed), nat.ptr(ws), ws.numel() * 8, nat.stream())
    return packed


def test_gram_config3_width_matvec_and_split_additivity():
    P, N = 16181, 4096
    Pp = _pad(P, 128)
    g = torch.Generator(device=DEV).manual_seed(7)
    JT = torch.zeros((Pp, N), dtype=torch.float64, device=DEV)
    JT[:P] = torch.randn((P, N), generator=g, dtype=torch.float64, device=DEV) + 0.25
    packed = _gram_full(JT, N, P, N)
    S = torch.empty((Pp, Pp), dtype=torch.float64, device=DEV)
    zero = torch.zeros(P, dtype=torch.float64, device=DEV)
    nat.call("vmcpde_gram_finalize", nat.ptr(packed), nat.ptr(zero), None, None, None, None, None, P, 1.0, 0.0, None, nat.ptr(S), Pp, None, 0,
             None, None, nat.stream())
    v = torch.randn(P, generator=g, dtype=torch.float64, device=DEV)
    ref = JT[:P] @ (JT[:P].T @ v)
    assert float(torch.linalg.norm(S[:P, :P] @ v - ref) / torch.linalg.norm(ref)) < 1e-13
    assert torch.equal(S[:P, :P], S[:P, :P].T)
    # sum over two sample ranges = the whole (the contraction is linear in the sample set); ragged split point on purpose
    n1 = 1500
    A1 = torch.zeros((Pp, _pad(n1, 16)), dtype=torch.float64, device=DEV)
    A1[:, :n1] = JT[:, :n1]
    A2 = JT[:, n1:].contiguous()
    n2 = N - n1
    ld2 = _pad(n2, 16)
    if ld2 != n2:
        B = torch.zeros((Pp, ld2), dtype=torch.float64, device=DEV)
        B[:, :n2] = A2
        A2 = B
    p1 = _gram_full(A1, A1.shape[1], P, n1)
    p2 = _gram_full(A2, A2.shape[1], P, n2)
    tiles = packed.view(-1, 128, 128)
    both = (p1 + p2).view(-1, 128, 128)
    nb = Pp // 128
    diag_ids = torch.tensor([i * (i + 1) // 2 + i for i in range(nb)], device=DEV)
    mask = torch.ones(tiles.shape[0], dtype=torch.bool, device=DEV)
    mask[diag_ids] = False
    scale = float(tiles[mask].abs().max())
    assert float((tiles[mask] - both[mask]).abs().max()) < 1e-12 * scale          # off-diagonal tiles: every entry is defined
    low = torch.tril(torch.ones(128, 128, dtype=torch.bool, device=DEV))            # diagonal tiles: lower triangle only
    assert float(((tiles[diag_ids] - both[diag_ids]) * low).abs().max()) < 1e-12 * scale


# ---- flow kernels of the two large nets against the autodiff oracle ---------------------------------------------------------
LARGE_NETS = [
    (2, 4, (43, 43), "Gauss", "no_add"),     # BASELINE C3 (P = 16 181): 140 KB of LDS per 64 samples, one workgroup per CU
    (6, 4, (19, 19), "Gauss", "no_add"),     # BASELINE C4 (P = 4155)
]


@pytest.mark.parametrize("cfg", LARGE_NETS)
def test_flow_parity_large_nets(cfg):
    from oracle import vmc_oracle as O
    from helpers import make_spec, random_theta, make_varstate, rel_err
    dim, depth, hidden, latent, coupling = cfg
    n = 40
    off = np.linspace(-0.5, 0.7, dim)
    spec = make_spec(dim, depth, hidden, latent=latent, coupling=coupling, offset=off, seed=11)
    th = random_theta(spec, 11)
    g = torch.Generator().manual_seed(11)
    z = torch.randn(1, n, dim, generator=g, dtype=torch.float64) * 0.8 + torch.as_tensor(off)
    vo = O.VarStateOracle(spec, th)
    vs = make_varstate(spec, th)
    assert vs.numParameters == {(43, 43): 16181, (19, 19): 4155}[hidden]
    x_o, lp_o = vo.flow_sample(z)
    x_p, lp_p = vs.flow_sample(z)
    assert rel_err(x_p, x_o) < 1e-12 and rel_err(lp_p, lp_o) < 1e-12
    lo, go, Jo = vo(x_o, "eval_coordgrads")
    lp, gp, Jp = vs(x_o, mode="eval_coordgrads")
    assert rel_err(lp, lo) < 1e-12 and rel_err(gp, go) < 1e-11
    assert float((Jp.cpu() - Jo).abs().max()) < 1e-11 * float(Jo.abs().max())
    Ho, Hp = vo.hessian(x_o), vs.hessian(x_o)
    assert float((Hp.cpu() - Ho).abs().max()) < 1e-10 * float(Ho.abs().max())


def test_tdvp_rhs_parity_config4_net():
    """One TDVP right-hand side of the C4 network (d = 6 Kramers problem, P = 4155) at N = 512 against the oracle."""
    import vmc_pde_amd as V
    from oracle import vmc_oracle as O
    from helpers import make_spec, random_theta, make_varstate, FixedSampler, rel_err
    dim, N, eq = 6, 512, "advection_hamiltonian_wDiss"
    spec = make_spec(dim, 4, (19, 19), offset=np.linspace(0.4, -0.3, dim), seed=3)
    th = random_theta(spec, 3, out_scale=0.03)
    g = torch.Generator().manual_seed(3)
    z = O.latent_from_normals(spec, th, torch.randn(N, dim, generator=g, dtype=torch.float64))
    eo, ep = O.EvolutionEquationOracle(dim, eq), V.EvolutionEquation(dim=dim, name=eq)
    eo.eqParams[eq]["params"]["lam"] = ep.eqParams[eq]["params"]["lam"] = 0.1
    # shifted system (tdvp.py:50-51): the spectral answer of the oracle and the on-device Cholesky solve the same matrix
    shift = 1e-4
    to = O.TDVPOracle(diagonalShift=shift)
    upd_o, _ = to.rhs(th, 0.1, O.VarStateOracle(spec, th), eo, z)
    vs = make_varstate(spec, th, sampler=FixedSampler(z.cuda()))
    tp = V.TDVP(diagonalShift=shift, solver="cholesky")
    upd_p, _ = tp(th.cuda(), 0.1, vs, ep, nSamplesTDVP=N, nSamplesObs=N, timings=None)
    assert rel_err(tp.S0, to.S0) < 1e-11 and rel_err(tp.F0, to.F0) < 1e-10
    assert int(tp.cholInfo.item()) == 0 and float(tp.solverResidual) < 1e-9
    S = to.S0 + torch.diag(shift * torch.diag(to.S0))
    exact = torch.linalg.solve(S + tp.svdTol * torch.diag(to.S0).max() * torch.eye(S.shape[0], dtype=S.dtype), to.F0)
    assert float(torch.linalg.norm(upd_p.cpu() - exact) / torch.linalg.norm(exact)) < 1e-8


@pytest.mark.parametrize("dim,hidden,N", [(2, (21, 21), 65536), (2, (43, 43), 65536), (6, (19, 19), 32768)])
def test_flow_round_trip_at_full_size(dim, hidden, N):
    """Size-independent properties at BASELINE's full sample counts (C2, C3, C4 shard): latent -> real (net.py:214-217) -> latent
    (net.py:209-213) returns the latent points; the log-density the sampling direction reports equals the one the evaluation
    direction computes at the same points (both are latent pdf minus log-Jacobian); every sample is finite."""
    import vmc_pde_amd as V
    ups = [[int(v) for v in np.random.default_rng(b).permutation(dim)[: dim // 2]] for b in range(4)]
    dns = [[i for i in range(dim) if i not in u] for u in ups]
    off = np.linspace(0.5, -0.5, dim)
    sampler = V.Sampler(key=5, dim=dim, name="Gauss", mcmc_info={"offset": off, "bound": 0.25})
    vs = V.VarState(sampler, dim, 1, 4, network_args=dict(intmediate=hidden, offset=off, latentSpaceName="Gauss", dim=dim),
                    inds_up=ups, inds_down=dns)
    vs.set_parameters(vs.net.init(2, out_scale=0.1))
    g = torch.Generator(device=DEV).manual_seed(N + dim)
    z = torch.randn((N, dim), generator=g, dtype=torch.float64, device=DEV) + torch.as_tensor(off, device=DEV)
    x, lp = vs.net.apply(vs.get_parameters(), z, evaluate=False, inv=True)
    z2, _ = vs.net.apply(vs.get_parameters(), x, evaluate=False, inv=False)
    assert bool(torch.isfinite(x).all()) and bool(torch.isfinite(lp).all())
    assert float((z2 - z).abs().max()) < 1e-11 * max(1.0, float(z.abs().max()))
    lp_eval = vs.net.apply(vs.get_parameters(), x, evaluate=True)
    scale = float(lp.abs().max())
    assert float((lp_eval - lp).abs().max()) < 1e-11 * scale
    # (the second value of the forward call is, as net.py:213 writes it, the latent pdf of the INPUT minus the log-Jacobian --
    # not a density of anything; pinned against the oracle in test_gpu_flow.py, not used here)


def test_spectral_solve_on_device_by_default_at_config4_width():
    """solver="eigh" (the reference's pseudo-inverse, tdvp.py:57-94) at P = 4155: diagonalizeOnDevice=None picks rocSOLVER from
    640 parameters on; host LAPACK (the reference's default) must give the same update, and both the oracle's."""
    import warnings
    import vmc_pde_amd as V
    from oracle import vmc_oracle as O
    from helpers import make_spec, random_theta, make_varstate, FixedSampler
    dim, N, eq = 6, 1024, "advection_hamiltonian_wDiss"
    spec = make_spec(dim, 4, (19, 19), offset=np.linspace(0.4, -0.3, dim), seed=3)
    th = random_theta(spec, 3, out_scale=0.03)
    g = torch.Generator().manual_seed(4)
    z = O.latent_from_normals(spec, th, torch.randn(N, dim, generator=g, dtype=torch.float64))
    eo, ep = O.EvolutionEquationOracle(dim, eq), V.EvolutionEquation(dim=dim, name=eq)
    to = O.TDVPOracle(svdTol=1e-8)
    upd_o, _ = to.rhs(th, 0.1, O.VarStateOracle(spec, th), eo, z)
    nF = torch.linalg.norm(to.F0)
    ups = {}
    for where in (None, False):
        vs = make_varstate(spec, th, sampler=FixedSampler(z.cuda()))
        tp = V.TDVP(svdTol=1e-8, diagonalizeOnDevice=where)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)       # (a non-converged rocSOLVER call falls back to the host)
            ups[where], _ = tp(th.cuda(), 0.1, vs, ep, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
        assert tp.ev is not None and tp.ev.numel() == vs.numParameters
        assert float(torch.linalg.norm(to.S0 @ (ups[where].cpu() - upd_o)) / nF) < 1e-9
    assert float(torch.linalg.norm(to.S0 @ (ups[None] - ups[False]).cpu()) / nF) < 1e-9


def _full_size_call(dim, hidden, N, eq, lam, jacobian_dtype=None, tdvp_kwargs=None, calls=2):
    import vmc_pde_amd as V
    ups = [[int(v) for v in np.random.default_rng(b).permutation(dim)[: dim // 2]] for b in range(4)]
    dns = [[i for i in range(dim) if i not in u] for u in ups]
    off = np.zeros(dim)
    sampler = V.Sampler(key=1, dim=dim, name="Gauss", mcmc_info={"offset": off, "bound": 0.25})
    vs = V.VarState(sampler, dim, 1, 4, network_args=dict(intmediate=hidden, offset=off, latentSpaceName="Gauss", dim=dim),
                    inds_up=ups, inds_down=dns, **({} if jacobian_dtype is None else {"jacobian_dtype": jacobian_dtype}))
    th = vs.net.init(1, out_scale=0.1)
    vs.set_parameters(th)
    ep = V.EvolutionEquation(dim=dim, name=eq)
    ep.eqParams[eq]["params"]["lam"] = lam
    tp = V.TDVP(**(dict(diagonalShift=1e-4, svdTol=1e-6, solver="cholesky") if tdvp_kwargs is None else tdvp_kwargs))
    out = []
    for _ in range(calls):      # second call: lagged centring + uncentred DMA Gram + fused moments (the steady state)
        upd, _ = tp(th, 0.0, vs, ep, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
        torch.cuda.synchronize()
        out.append((upd.clone(), float(tp.solverResidual), int(tp.cholInfo.item()) if tp.cholInfo is not None else 0, float(tp.tdvp_error)))
    return vs, tp, out


def test_full_size_config3_tdvp_call():
    """BASELINE config 3 at full size: P = 16 181, 65 536 samples (J = 8.5 GB, S = 2.1 GB) -- sample, flow, Gram, Cholesky."""
    vs, tp, out = _full_size_call(2, (43, 43), 65536, "advection_hamiltonian_wDiss", 0.1)
    assert vs.numParameters == 16181
    for upd, resid, info, err in out:
        assert torch.isfinite(upd).all() and info == 0 and resid < 1e-8 and 0.0 <= err < 1.0
    assert tp._uncentred_ok       # the second call ran the steady-state path
    # at this size the regularised solve goes through the pivoted low-rank factor of S (Woodbury + refinement against the full matrix)
    ci = tp.choleskyInfo             # (the first call plans with a host read of the rank, the following ones run the plan)
    assert ci["engine"] == "lowrank" and ci["mode"] == "static" and ci["planned_rank"] < 768 and ci["rmax"] <= 768, ci
    # same samples are redrawn with a new stream id, so the two updates agree to Monte-Carlo accuracy only
    assert float(torch.linalg.norm(out[0][0] - out[1][0]) / torch.linalg.norm(out[0][0])) < 0.5
    # S v = J^T (J v) / N - mean (mean . v) on the retained Jacobian of the last call
    z = tp._lazy
    P, n = z["P"], z["n"]
    g = torch.Generator(device=DEV).manual_seed(3)
    v = torch.randn(P, generator=g, dtype=torch.float64, device=DEV)
    J = z["JT"][:P, :n]
    ref = J @ (J.T @ v) / n - z["mean"] * (z["mean"] @ v)
    assert float(torch.linalg.norm(tp.S0 @ v - ref) / torch.linalg.norm(ref)) < 1e-10


def test_full_size_config4_shard_tdvp_call():
    """BASELINE config 4, one GPU's shard: d = 6, hidden (19,19), P = 4155, 32 768 samples."""
    vs, tp, out = _full_size_call(6, (19, 19), 32768, "advection_hamiltonian_wDiss", 0.0)
    assert vs.numParameters == 4155
    for upd, resid, info, err in out:
        assert torch.isfinite(upd).all() and info == 0 and resid < 1e-8 and 0.0 <= err < 1.0
    assert tp._uncentred_ok


def test_full_size_config5_call_on_the_dma_staged_fp32_gram():
    """BASELINE config 5 at full size (C2 with an fp32 Jacobian, fp64 products and sums): the steady-state call takes the DMA-staged Gram over
    sample pairs (csrc/gram.hip::gram_w32_kernel); S0 v = J^T (J v) / N - m (m . v) on the stored fp32 Jacobian (widened), the same step
    with the register-staged loop (VMCPDE_GRAM_F32_DMA=0) gives the same S to a few sqrt(N) ulp and the same update to 1e-9 |F| as the
    matrix sees it, and the step is reproducible bit for bit."""
    import os
    runs = {}
    for mode in ("1", "0", "1"):
        os.environ["VMCPDE_GRAM_F32_DMA"] = mode
        try:
            vs, tp, out = _full_size_call(2, (21, 21), 65536, "advection_hamiltonian_wDiss", 0.1, jacobian_dtype=torch.float32)
        finally:
            os.environ.pop("VMCPDE_GRAM_F32_DMA", None)
        assert vs.numParameters == 4213 and tp._uncentred_ok
        for upd, resid, info, err in out:
            assert torch.isfinite(upd).all() and info == 0 and resid < 1e-8 and 0.0 <= err < 1.0
        runs.setdefault(mode, []).append((out[-1][0], tp.S0.clone(), tp.F0.clone()))
        if mode == "1" and len(runs["1"]) == 1:
            z = tp._lazy
            P, n = z["P"], z["n"]
            assert z["JT"].dtype == torch.float32
            g = torch.Generator(device=DEV).manual_seed(3)
            v = torch.randn(P, generator=g, dtype=torch.float64, device=DEV)
            J = z["JT"][:P, :n].double()
            ref = J @ (J.T @ v) / n - z["mean"] * (z["mean"] @ v)
            assert float(torch.linalg.norm(tp.S0 @ v - ref) / torch.linalg.norm(ref)) < 1e-10
        del vs, tp
    (u1, S1, F1), (u1b, S1b, F1b) = runs["1"]
    u0, S0, F0 = runs["0"][0]
    assert torch.equal(u1, u1b) and torch.equal(S1, S1b)                              # reproducible
    assert torch.equal(F1, F0) and not torch.equal(S1, S0)                           # (the same moments pass; another Gram loop)
    d = torch.sqrt(torch.diagonal(S0).clamp_min(0))
    scale = torch.outer(d, d).clamp_min(1e-300)
    assert float(((S1 - S0).abs() / scale).max()) < 4e-16 * 256 + 1e-13             # (a few sqrt(65 536) ulp of sqrt(S_pp S_qq); centring cancels a little)
    nF = float(torch.linalg.norm(F0))
    assert float(torch.linalg.norm(S0 @ (u1 - u0))) < 1e-9 * nF


def test_full_size_reference_default_call_is_reproducible_and_consistent():
    """`TDVP()` exactly as the reference constructs it at the headline size (P = 4213, 65 536 samples): the rank-revealing engine with the own
    tridiagonalisation + stedc + back-transformation core and the batched Householder chain.  Two independent runs give the same (ev, update)
    bit for bit (no atomics anywhere: every rank of a sharded run must form the same update); S V = V diag(ev) on the Ritz modes; snr from
    the Ritz vectors is finite, zero exactly on the null modes."""
    runs = []
    for _ in range(2):
        vs, tp, out = _full_size_call(2, (21, 21), 65536, "advection_hamiltonian_wDiss", 0.1, tdvp_kwargs={})
        assert tp.spectralInfo["engine"] == "lowrank" and tp.spectralInfo["core"].startswith("sytrd(own)"), tp.spectralInfo
        snr = tp.snr
        runs.append((out[-1][0], tp.ev.clone(), snr.clone()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])
    ev, V = tp.ev, tp.V
    ritz = ev != 0
    lmax = float(ev[-1])
    r = int(ritz.sum())
    assert 100 < r <= 256 and bool((ev[1:] >= ev[:-1]).all())
    Vr = V[:, ritz]
    assert float(torch.linalg.norm(tp.S0 @ Vr - Vr * ev[ritz])) < 1e-11 * lmax * np.sqrt(r)
    assert float((Vr.T @ Vr - torch.eye(r, dtype=torch.float64, device=DEV)).abs().max()) < 1e-13
    assert bool(torch.isfinite(snr).all()) and not snr[~ritz].any() and int((snr[ritz] > 0).sum()) >= r - 2


def test_full_size_step_is_bitwise_reproducible():
    """The whole step at the C4 shard size and at C2 (sample, flow, Gram with split reduction, Cholesky, substitution): a second,
    independent run from the same seeds gives the same update bit for bit -- no atomics or arrival-order dependent sums anywhere
    (the replicated solve of the multi-GPU scheme relies on it: theta is never broadcast after step 0)."""
    for dim, hidden, N, lam in ((6, (19, 19), 32768, 0.0), (2, (21, 21), 65536, 0.1)):
        runs = [_full_size_call(dim, hidden, N, "advection_hamiltonian_wDiss", lam)[2] for _ in range(2)]
        for (u0, r0, i0, e0), (u1, r1, i1, e1) in zip(*runs):
            assert torch.equal(u0, u1) and r0 == r1 and e0 == e1 and i0 == i1 == 0


def test_two_host_threads_on_their_own_streams():
    """include/vmcpde.h: thread-compatible, no process-wide mutable state.  Two host threads factor different matrices and build
    different Gram matrices at the same time, each on its own stream with its own workspaces; results must equal the serial ones
    bit for bit (the per-GPU context -- permutation table, attribute flags -- is created under a lock and read-only afterwards)."""
    import threading
    P, N = 1500, 4096
    mats = [_spd_gpu(P, 100 + k) for k in range(2)]
    g = torch.Generator(device=DEV).manual_seed(5)
    jts = [torch.randn((_pad(P, 128), N), generator=g, dtype=torch.float64, device=DEV) for _ in range(2)]
    for jt in jts:
        jt[P:] = 0.0

    def work(k, stream, out):
        with torch.cuda.stream(stream):
            for _ in range(3):
                Ap, *_ = _factor(mats[k], P)
                npk = nat.lib().vmcpde_gram_packed_doubles(P)
                packed = torch.empty(npk, dtype=torch.float64, device=DEV)
                need = nat.lib().vmcpde_gram_workspace_bytes(P, N)
                ws = torch.empty(need // 8, dtype=torch.float64, device=DEV)
                nat.call("vmcpde_gram", nat.ptr(jts[k]), N, 0, P, N, None, None, nat.ptr(packed), nat.ptr(ws), need, nat.stream())
            stream.synchronize()
            out[k] = (Ap[:P, :P].tril().clone(), packed.clone())

    serial, threaded = {}, {}
    for k in range(2):
        work(k, torch.cuda.current_stream(), serial)
    streams = [torch.cuda.Stream() for _ in range(2)]
    torch.cuda.synchronize()
    ts = [threading.Thread(target=work, args=(k, streams[k], threaded)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for k in range(2):
        assert torch.equal(serial[k][0], threaded[k][0]) and torch.equal(serial[k][1], threaded[k][1])
