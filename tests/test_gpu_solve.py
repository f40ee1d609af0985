"""GPU parity: blocked Cholesky, triangular solves, mat-vecs and the spectral cut-off coefficients (csrc/solve.hip)."""
import numpy as np
import pytest
import torch

from vmc_pde_amd import _native as nat

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _pad(v, m):
    return (v + m - 1) // m * m


def _spd(P, seed, cond=1e6):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((P, P)))
    ev = np.logspace(0, -np.log10(cond), P)
    return (Q * ev) @ Q.T


def _chol(Amat, P):
    Pp = _pad(P, 128)
    A = torch.zeros((Pp, Pp), dtype=torch.float64, device=DEV)
    A[:P, :P] = torch.as_tensor(Amat)
    need = nat.lib().vmcpde_chol_workspace_bytes(P)
    ws = torch.empty(need // 8, dtype=torch.float64, device=DEV)
    info = torch.zeros(1, dtype=torch.int32, device=DEV)
    nat.call("vmcpde_cholesky", nat.ptr(A), Pp, P, nat.ptr(info), nat.ptr(ws), need, nat.stream())
    return A, Pp, ws, need, info


@pytest.mark.parametrize("P", [5, 128, 129, 300, 1000])
def test_cholesky_and_solve_parity(P):
    Amat = _spd(P, P, cond=1e4)
    A, Pp, ws, need, info = _chol(Amat, P)
    L = np.tril(A.cpu().numpy()[:P, :P])
    Lref = np.linalg.cholesky(Amat)
    assert int(info.item()) == 0
    assert np.abs(L - Lref).max() < 1e-10 * np.abs(Lref).max()
    rng = np.random.default_rng(P + 1)
    F = rng.standard_normal(P)
    u = torch.empty(P, dtype=torch.float64, device=DEV)
    F_d = torch.as_tensor(F).to(DEV)
    nat.call("vmcpde_chol_solve", nat.ptr(A), Pp, P, nat.ptr(F_d), nat.ptr(u), nat.ptr(ws), need, nat.stream())
    uref = np.linalg.solve(Amat, F)
    assert np.linalg.norm(Amat @ u.cpu().numpy() - F) / np.linalg.norm(F) < 1e-9
    assert np.linalg.norm(u.cpu().numpy() - uref) / np.linalg.norm(uref) < 1e-8


def test_cholesky_pins_dead_parameters():
    # H7: a parameter with zero gradient gives a zero row/column of S; its update component must be 0
    P = 200
    Amat = _spd(P, 3, cond=1e3)
    dead = [7, 130, 199]
    Amat[dead, :] = 0.0
    Amat[:, dead] = 0.0
    A, Pp, ws, need, info = _chol(Amat, P)
    assert int(info.item()) == len(dead)
    F = np.random.default_rng(2).standard_normal(P)
    F[dead] = 0.0
    u = torch.empty(P, dtype=torch.float64, device=DEV)
    F_d = torch.as_tensor(F).to(DEV)
    nat.call("vmcpde_chol_solve", nat.ptr(A), Pp, P, nat.ptr(F_d), nat.ptr(u), nat.ptr(ws), need, nat.stream())
    un = u.cpu().numpy()
    live = [i for i in range(P) if i not in dead]
    assert np.all(un[dead] == 0.0)
    assert np.allclose(un[live], np.linalg.solve(Amat[np.ix_(live, live)], F[live]), rtol=1e-9)


def test_gemv_and_pinv_coeffs():
    rng = np.random.default_rng(0)
    R, Cc = 300, 411
    Amat = rng.standard_normal((R, Cc))
    A = torch.as_tensor(Amat).to(DEV).contiguous()
    x = rng.standard_normal(Cc)
    y = torch.empty(R, dtype=torch.float64, device=DEV)
    x_d = torch.as_tensor(x).to(DEV)
    nat.call("vmcpde_gemv", nat.ptr(A), Cc, R, Cc, 0, nat.ptr(x_d), nat.ptr(y), nat.stream())
    assert np.allclose(y.cpu().numpy(), Amat @ x, rtol=1e-12, atol=1e-12)
    xt = rng.standard_normal(R)
    yt = torch.empty(Cc, dtype=torch.float64, device=DEV)
    xt_d = torch.as_tensor(xt).to(DEV)
    nat.call("vmcpde_gemv", nat.ptr(A), Cc, R, Cc, 1, nat.ptr(xt_d), nat.ptr(yt), nat.stream())
    assert np.allclose(yt.cpu().numpy(), Amat.T @ xt, rtol=1e-12, atol=1e-12)
    # tdvp.py:82-89 incl. the hard 1e-14 cut, a negative round-off eigenvalue and the SNR factor
    P = 64
    ev = np.sort(np.concatenate([[-4e-14, 1e-16, 1e-13, 3e-12], np.logspace(-11, 2, P - 4)]))
    VtF, snr = rng.standard_normal(P), np.abs(rng.standard_normal(P)) * 3
    rel = np.abs(ev / ev[-1])
    inv = np.where(rel > 1e-14, 1.0 / ev, 0.0)
    reg = 1.0 / (1.0 + (1e-11 / rel) ** 6)
    for use_snr in (False, True):
        r = reg * (1.0 / (1.0 + (2.0 / snr) ** 6)) if use_snr else reg
        coef = torch.empty(P, dtype=torch.float64, device=DEV)
        ev_d, VtF_d, snr_d = (torch.as_tensor(a).to(DEV) for a in (ev, VtF, snr))   # keep alive across the launch
        nat.call("vmcpde_pinv_coeffs", nat.ptr(ev_d), nat.ptr(VtF_d), nat.ptr(snr_d) if use_snr else None, P, 1e-11, 2.0,
                 nat.ptr(coef), nat.stream())
        assert np.allclose(coef.cpu().numpy(), inv * r * VtF, rtol=1e-13, atol=0)


@pytest.mark.parametrize("P", [5, 127, 129, 300, 1000])
def test_bordered_factorisation_carries_the_forward_substitution(P):
    # Cholesky of [[A, F], [F^T, beta]]: the last row of the factor is y = L^-1 F, so u = L^-T y needs only the back
    # substitution (vmcpde_chol_backsolve); must agree with factor + vmcpde_chol_solve and with a dense solve
    Amat = _spd(P, P + 7, cond=1e5)
    rng = np.random.default_rng(P + 2)
    F = rng.standard_normal(P)
    Pp = _pad(P + 1, 128)
    A = torch.zeros((Pp, Pp), dtype=torch.float64, device=DEV)
    A[:P, :P] = torch.as_tensor(Amat)
    A[P, :P] = torch.as_tensor(F)
    A[P, P] = 1e200
    need = nat.lib().vmcpde_chol_workspace_bytes(P + 1)
    ws = torch.empty(need // 8, dtype=torch.float64, device=DEV)
    info = torch.zeros(1, dtype=torch.int32, device=DEV)
    nat.call("vmcpde_cholesky", nat.ptr(A), Pp, P + 1, nat.ptr(info), nat.ptr(ws), need, nat.stream())
    assert int(info.item()) == 0
    Lref = np.linalg.cholesky(Amat)
    yref = np.linalg.solve(Lref, F)
    y = A[P, :P].cpu().numpy()
    assert np.linalg.norm(y - yref) / np.linalg.norm(yref) < 1e-9
    u = torch.empty(P, dtype=torch.float64, device=DEV)
    nat.call("vmcpde_chol_backsolve", nat.ptr(A), Pp, P, nat.ptr(A[P]), nat.ptr(u), nat.ptr(ws), need, nat.stream())
    uref = np.linalg.solve(Amat, F)
    assert np.linalg.norm(Amat @ u.cpu().numpy() - F) / np.linalg.norm(F) < 1e-9
    assert np.linalg.norm(u.cpu().numpy() - uref) / np.linalg.norm(uref) < 1e-7
    # the two-call path on the same matrix
    A2, Pp2, ws2, need2, _ = _chol(Amat, P)
    u2 = torch.empty(P, dtype=torch.float64, device=DEV)
    F_d = torch.as_tensor(F).to(DEV)
    nat.call("vmcpde_chol_solve", nat.ptr(A2), Pp2, P, nat.ptr(F_d), nat.ptr(u2), nat.ptr(ws2), need2, nat.stream())
    assert np.linalg.norm((u - u2).cpu().numpy()) / np.linalg.norm(uref) < 1e-9


@pytest.mark.parametrize("P", [1, 5, 128, 129, 206, 256, 300, 384, 500, 512])
def test_small_systems_substitute_in_one_launch(P, monkeypatch):
    """Round 6: both substitutions of a system of <= 4 blocks (the r x r matrix G of the Woodbury application) run as ONE workgroup with the right-hand
    side in LDS -- the arithmetic of the block-row chain in the same order, so the same bits (VMCPDE_SOLVE_SMALL=0), in place or not; and the right answer."""
    from vmc_pde_amd import _native as nat
    Ppad = (P + 127) // 128 * 128
    g = torch.Generator().manual_seed(P)
    M = torch.randn(P, 2 * P + 3, generator=g, dtype=torch.float64)
    A = torch.zeros((Ppad, Ppad), dtype=torch.float64)
    A[:P, :P] = M @ M.T / (2 * P + 3) + 0.1 * torch.eye(P, dtype=torch.float64)
    A[range(P, Ppad), range(P, Ppad)] = 1.0
    F = torch.randn(P, generator=g, dtype=torch.float64)
    Ad, Fd = A.cuda(), F.cuda()
    need = nat.lib().vmcpde_chol_workspace_bytes(P)
    ws = torch.empty(need // 8, dtype=torch.float64, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    nat.call("vmcpde_cholesky", nat.ptr(Ad), Ppad, P, nat.ptr(info), nat.ptr(ws), need, nat.stream())
    outs = []
    for mode in ("1", "0"):
        monkeypatch.setenv("VMCPDE_SOLVE_SMALL", mode)
        u = torch.full((P,), float("nan"), dtype=torch.float64, device="cuda")
        nat.call("vmcpde_chol_solve", nat.ptr(Ad), Ppad, P, nat.ptr(Fd), nat.ptr(u), nat.ptr(ws), need, nat.stream())
        v = Fd.clone()
        nat.call("vmcpde_chol_solve", nat.ptr(Ad), Ppad, P, nat.ptr(v), nat.ptr(v), nat.ptr(ws), need, nat.stream())
        assert torch.equal(u, v)
        outs.append(u.cpu())
    ref = torch.linalg.solve(A[:P, :P], F)
    assert torch.equal(outs[0], outs[1]) and float((outs[0] - ref).abs().max() / ref.abs().max()) < 1e-10
