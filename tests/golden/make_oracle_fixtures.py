"""Generate tests/golden/oracle_vectors.npz and oracle_heun_step.npz from the CPU oracle.

    python tests/golden/make_oracle_fixtures.py

The reference itself (JAX/Flax) cannot run in this image, so these vectors are produced by oracle/vmc_oracle.py --
the restatement of the reference's algorithm -- with fixed seeds on small shapes: d in {2,3,6,8}, depth in {1,4},
hidden (3,) and (4,4), the four coupling variants, both latent densities, N = 64.  Per case the file holds the inputs
(flat theta, partitions, offset, latent draw z, t) and the outputs (x, logp, grad_x, Hessian, J, Eloc per equation, S0, F0,
SExp (in full for P <= 120, else diagonal + one matrix-vector probe), ev, update, residual, tdvp_error).  They pin the oracle against silent edits (CPU test) and give the HIP path a
fixed target that does not depend on the oracle being importable (GPU test).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import vmc_oracle as O          # noqa: E402
from helpers import make_spec, random_theta  # noqa: E402

N = 64
T_EVAL = 0.3
# (tag, dim, depth, hidden, latent, coupling, equations)
CASES = [
    ("d2_b4_h1_gauss_noadd", 2, 4, (1,), "Gauss", "no_add", ["diffusion", "diffusion_drift", "advection_hamiltonian_wDiss"]),
    ("d2_b1_h44_gauss_plain", 2, 1, (4, 4), "Gauss", "plain", ["advection_paper", "advection_hamiltonian"]),
    ("d3_b4_h3_gauss_jaceq1", 3, 4, (3,), "Gauss", "jac_eq_1", ["diffusion", "diffusion_anisotropic"]),
    ("d6_b4_h3_gauss_diffadd", 6, 4, (3,), "Gauss", "different_add", ["advection_hamiltonian_wDiss", "diffusion_anisotropic"]),
    ("d8_b1_h3_student_noadd", 8, 1, (3,), "Student_t", "no_add", ["diffusion"]),
    ("d8_b4_h44_student_noadd", 8, 4, (4, 4), "Student_t", "no_add", ["diffusion", "diffusion_drift"]),
    ("d3_b2_h3_gauss_noadd_global", 3, 2, (3,), "Gauss", "no_add+global", ["diffusion"]),      # global_change, net.py:72
]


FULL_MATRIX_MAX_P = 120


def probe_vector(P):
    return torch.cos(0.37 * torch.arange(P, dtype=torch.float64) + 0.1)


def case_inputs(tag, dim, depth, hidden, latent, coupling, seed):
    off = np.linspace(0.4, -0.3, dim)
    spec = make_spec(dim, depth, hidden, latent=latent, coupling=coupling, offset=off, seed=seed)
    th = random_theta(spec, seed, out_scale=0.04)
    g = torch.Generator().manual_seed(100 + seed)
    nrm = torch.randn(N, dim, generator=g, dtype=torch.float64)
    chi2 = None
    if latent == "Student_t":
        nu = float(np.exp(float(O.unflatten(spec, th)["dist_params"][0])) + 1.0)
        chi2 = torch.from_numpy(np.random.default_rng(200 + seed).chisquare(nu, N))
    z = O.latent_from_normals(spec, th, nrm, chi2)
    return spec, th, z


def make_eq(dim, name):
    eo = O.EvolutionEquationOracle(dim, name)
    if "hamiltonian" in name:
        eo.eqParams[name]["params"]["lam"] = 0.1
    return eo


def main():
    out = {}
    for k, (tag, dim, depth, hidden, latent, coupling, eqs) in enumerate(CASES):
        spec, th, z = case_inputs(tag, dim, depth, hidden, latent, coupling, seed=k + 1)
        vo = O.VarStateOracle(spec, th)
        x, logp_s = vo.flow_sample(z)
        logp, g, J = vo(x, mode="eval_coordgrads")
        H = vo.hessian(x)
        out[f"{tag}/theta"] = th.numpy()
        out[f"{tag}/inds_up"] = np.asarray(spec.inds_up, dtype=np.int32)
        out[f"{tag}/inds_down"] = np.asarray(spec.inds_down, dtype=np.int32)
        out[f"{tag}/offset"] = np.asarray(spec.offset)
        out[f"{tag}/z"] = z.numpy()
        out[f"{tag}/x"] = x.numpy()
        out[f"{tag}/logp_sample"] = logp_s.numpy()
        out[f"{tag}/logp"] = logp.numpy()
        out[f"{tag}/grad_x"] = g.numpy()
        out[f"{tag}/hessian"] = H.numpy()
        out[f"{tag}/J"] = J.numpy()
        for name in eqs:
            eo = make_eq(dim, name)
            to = O.TDVPOracle()
            upd, aux = to.rhs(th, T_EVAL, vo, eo, z)
            pre = f"{tag}/{name}"
            out[f"{pre}/Eloc"] = aux["Eloc"].numpy()
            if name == "diffusion_anisotropic":
                out[f"{pre}/D"] = eo.eqParams[name]["D"].numpy()
            for key in ("F0", "ev"):
                out[f"{pre}/{key}"] = getattr(to, key).numpy()
            for key in ("S0", "SExp"):     # full matrices for small P, else diagonal + one probe product (keeps the file small)
                M = getattr(to, key)
                if M.shape[0] <= FULL_MATRIX_MAX_P:
                    out[f"{pre}/{key}"] = M.numpy()
                else:
                    out[f"{pre}/{key}_diag"] = torch.diag(M).numpy()
                    out[f"{pre}/{key}_probe"] = (M @ probe_vector(M.shape[0])).numpy()
            out[f"{pre}/update"] = upd.numpy()
            out[f"{pre}/residual"] = np.float64(to.solverResidual)
            out[f"{pre}/tdvp_error"] = np.float64(to.tdvp_error)
            out[f"{pre}/Eloc_stats"] = np.asarray([float(to.ElocMean), float(to.ElocVar), float(to.ElocMeanAbs)])
    np.savez_compressed(os.path.join(HERE, "oracle_vectors.npz"), **out)
    print("oracle_vectors.npz:", len(out), "arrays,", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")

    # one full Heun step (stepper.py:131-139) of the mwe-like net, latent draws injected for both stages
    tag, dim, depth, hidden, latent, coupling, _ = CASES[0]
    spec, th0, z0 = case_inputs(tag, dim, depth, hidden, latent, coupling, seed=1)
    g = torch.Generator().manual_seed(991)
    nrm = [torch.randn(512, dim, generator=g, dtype=torch.float64) for _ in range(2)]
    vo = O.VarStateOracle(spec, th0)
    eo = make_eq(dim, "diffusion")
    to = O.TDVPOracle()
    zs, stage_in, stage_out, stage_F = [], [], [], []

    def f(y, t, k):
        z = O.latent_from_normals(spec, y, nrm[k])       # the latent draw depends on theta (mu, L) -- sampler.py:86
        zs.append(z.numpy())
        u, _ = to.rhs(y, t, vo, eo, z)
        stage_in.append(y.numpy().copy())                # the state each stage is evaluated at, its update and its force
        stage_out.append(u.numpy().copy())
        stage_F.append(to.F0.numpy().copy())
        return u, None

    dt = O.fixed_stepper_dt(1e-5, 1.3, 1e-4)       # FixedStepper(timeStep=1e-5, increase_fac=1.3, maxStep=1e-4)
    th1, _ = O.heun_step(f, th0, 0.0, dt)
    np.savez_compressed(os.path.join(HERE, "oracle_heun_step.npz"), theta0=th0.numpy(), theta1=th1.numpy(), dt=np.float64(dt),
                        normals0=nrm[0].numpy(), normals1=nrm[1].numpy(), z0=zs[0], z1=zs[1],
                        stage_theta=np.stack(stage_in), stage_update=np.stack(stage_out), stage_F=np.stack(stage_F),
                        inds_up=np.asarray(spec.inds_up, dtype=np.int32), inds_down=np.asarray(spec.inds_down, dtype=np.int32),
                        offset=np.asarray(spec.offset))
    print("oracle_heun_step.npz: |theta1-theta0| =", float(torch.linalg.norm(th1 - th0)))


if __name__ == "__main__":
    torch.set_num_threads(4)
    main()
