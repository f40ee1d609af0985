"""Diagnostic: eigh-path update parity split into (a) solve parity on identical (S, F) and (b) end-to-end agreement restricted to
the modes with lambda > 1e-9 lambda_max, next to the a-priori perturbation bound from the measured S / F discrepancy."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import vmc_oracle as O
from helpers import make_varstate, FixedSampler
import test_gpu_tdvp as T
import vmc_pde_amd as V

torch.set_num_threads(8)
for case in T.CASES:
    dim, depth, hidden, latent, coupling, eq, N = case
    spec, th, nrm, chi2, eo, ep = T._problem(dim, depth, hidden, latent, coupling, eq, N, lam=0.1)
    z = O.latent_from_normals(spec, th, nrm, chi2)
    for useSNR in (False, True):
        to = O.TDVPOracle(useSNR=useSNR)
        upd_o, aux = to.rhs(th, 0.2, O.VarStateOracle(spec, th), eo, z)
        vs = make_varstate(spec, th, sampler=FixedSampler(z.cuda()))
        tp = V.TDVP(useSNR=useSNR)
        upd_p, info = tp(th.cuda(), 0.2, vs, ep, nSamplesTDVP=N, nSamplesObs=N, timings=None)
        up = upd_p.cpu()
        nF = torch.linalg.norm(to.F0)
        Sp, Fp = tp.S0.cpu(), tp.F0.cpu()
        # (a) oracle solve on the product's S, F
        t2 = O.TDVPOracle(useSNR=False)
        ev, Vv = np.linalg.eigh(Sp.numpy()); ev, Vv = torch.from_numpy(ev), torch.from_numpy(Vv)
        rel = torch.abs(ev / ev[-1]); inv = torch.where(rel > 1e-14, 1.0 / ev, torch.zeros_like(ev)); reg = 1 / (1 + (1e-11 / rel) ** 6)
        if useSNR:
            reg = reg * (1 / (1 + (2.0 / tp.snr.cpu()) ** 6))
        u2 = Vv @ (inv * reg * (Vv.T @ Fp))
        a_upd = float(torch.linalg.norm(Sp @ (up - u2)) / nF)
        msq = float((aux["Eloc"] ** 2).mean())
        err = lambda u, S, F: float(1 + (u @ S @ u - 2 * F @ u) / msq)
        a_err = abs(err(up, Sp, Fp) - err(u2, Sp, Fp))
        # (b) well-conditioned subspace of the oracle
        big = to.ev > 1e-9 * to.ev[-1]
        W = to.V[:, big]
        proj = lambda u: W @ (W.T @ u)
        b_upd = float(torch.linalg.norm(to.S0 @ proj(up - upd_o)) / nF)
        b_err = abs(err(proj(up), to.S0, to.F0) - err(proj(upd_o), to.S0, to.F0))
        full_upd = float(torch.linalg.norm(to.S0 @ (up - upd_o)) / nF)
        dS = float(torch.linalg.matrix_norm(Sp - to.S0, 2)); dF = float(torch.linalg.norm(Fp - to.F0))
        bound = (dS * float(torch.linalg.norm(proj(upd_o))) + dF) / float(nF)
        snr_err = None
        if True:
            sp, so = tp.snr.cpu(), to.snr
            snr_err = float(((sp[big] - so[big]).abs() / so[big].abs()).max())
        print(f"{case} SNR={useSNR}: P={Sp.shape[0]} nbig={int(big.sum())} (a) upd {a_upd:.2e} err {a_err:.2e} | (b) upd {b_upd:.2e} err {b_err:.2e} | full {full_upd:.2e} "
              f"| dS2 {dS:.2e} |u_w| {float(torch.linalg.norm(proj(upd_o))):.2e} |F| {float(nF):.2e} bound {bound:.2e} | snr rel err (big) {snr_err:.2e} "
              f"| d tdvp_err {abs(float(tp.tdvp_error)-float(to.tdvp_error)):.2e}", flush=True)
