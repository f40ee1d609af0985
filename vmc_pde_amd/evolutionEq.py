"""PDE right-hand sides d_t log p (reference: vmc_fluids/evolutionEq.py).

`EvolutionEquation(dim, name)(vState, configs, t) -> (Eloc (1,N), param_grads (1,N,P), logProbs (1,N))` as
evolutionEq.py:81-119.  The reference evaluates the network twice (value_and_grad, then a full jacrev(jacfwd)
Hessian pass) and contracts the Hessian afterwards; every equation here only needs sum_k v_k^T H v_k over at most d
directions, which the flow kernel produces in the same pass as the Jacobian (csrc/flow.hip), and the small epilogue
kernel `vmcpde_rhs` combines them.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _native as nat
from . import global_defs


def _getRandomD_matrix(dim, seed=0):
    """evolutionEq.py:18-20: A^T A, A ~ N(0,1) (seeded torch generator instead of the JAX key)."""
    g = torch.Generator().manual_seed(seed)
    A = torch.randn((dim, dim), generator=g, dtype=torch.float64)
    return A.T @ A


@dataclass
class EvolutionEquation:
    dim: int = 2
    name: str = "diffusion"

    def __post_init__(self):
        # evolutionEq.py:61-77 (public: read by visualization.py:132-133)
        self.eqParams = {"diffusion": {"D": 1},
                         "diffusion_anisotropic": {"D": _getRandomD_matrix(self.dim)},
                         "diffusion_drift": {"D": 1, "mu": 4},
                         "advection_paper": {"params": {"T": 5}, "vel_field": "MLPaper"},
                         "advection_hamiltonian": {"params": {"m": 1.0, "omega": 1.0, "lam": 0.0}, "vel_field": "hamiltonian"},
                         "advection_hamiltonian_wDiss": {"params": {"m": 1.0, "omega": 1.0, "T": 10.0, "gamma": 1.0, "lam": 0.0},
                                                         "vel_field": "hamiltonian"}}
        if self.name not in self.eqParams:
            raise KeyError(self.name)      # evolutionEq.py:54-60 dispatch table

    def _consts_and_dirs(self, device):
        c = nat.EqConsts()
        c.kind = nat.EQ_KIND[self.name]
        e = self.eqParams[self.name]
        d = self.dim
        dirs, Dmat = None, None
        if self.name in ("diffusion", "diffusion_drift"):
            c.D = float(e["D"])
            c.mu = float(e.get("mu", 0.0))
            dirs = np.eye(d)                                   # tr H
        elif self.name == "diffusion_anisotropic":
            Dm = torch.as_tensor(e["D"], dtype=torch.float64).cpu()
            Dm = 0.5 * (Dm + Dm.T)                             # 'abij,ji' only sees the symmetric part of D
            evals, evecs = torch.linalg.eigh(Dm)
            # tr(H D) = sum_k lam_k v_k^T H v_k ; fold |lam_k|^(1/2) into the direction, the sign into the order below
            if bool((evals < -1e-14 * evals.abs().max()).any()):
                raise ValueError("diffusion matrix must be positive semi-definite")
            dirs = (evecs * torch.sqrt(evals.clamp_min(0.0))).T.numpy()
            Dmat = Dm.contiguous().to(device)
        elif self.name == "advection_paper":
            c.T = float(e["params"]["T"])
        else:
            p = e["params"]
            c.m, c.omega, c.lam = float(p["m"]), float(p["omega"]), float(p["lam"])
            c.coupled = int(bool(p.get("coupled", False)))     # evolutionEq.py:33-38; the reference never passes True
            if self.name == "advection_hamiltonian_wDiss":
                c.T, c.gamma = float(p["T"]), float(p["gamma"])
                dirs = np.eye(d)[1::2]                         # momentum axes, evolutionEq.py:114
        return c, dirs, Dmat

    def __call__(self, vState, configs, t):
        """evolutionEq.py:81-82."""
        c, dirs, Dmat = self._consts_and_dirs(vState.device)
        r = vState.evaluate(configs, dirs=dirs)
        return self._rhs(vState, r, c, Dmat, t)

    def from_latent(self, vState, latent_space_samples, t):
        """The same for a batch given by its LATENT points (what `vState.sampler` draws): one launch produces the points x themselves (the inverse
        flow, var_state.py:79), log p, grad_x, the directional second derivatives and the Jacobian (`VarState.evaluate_from_latent`).
        Returns (configs (1,N,d), Eloc (1,N), param_grads (1,N,P), logProbs (1,N))."""
        c, dirs, Dmat = self._consts_and_dirs(vState.device)
        r = vState.evaluate_from_latent(latent_space_samples, dirs=dirs)
        return (r["x"][None, ...], *self._rhs(vState, r, c, Dmat, t))

    def _rhs(self, vState, r, c, Dmat, t):
        n = r["n"]
        eloc = torch.empty(n, dtype=global_defs.tReal, device=vState.device)
        hv = r["hv"]
        with torch.cuda.device(vState.device):
            nat.call("vmcpde_rhs", C.byref(c), self.dim, nat.ptr(r["x"]), nat.ptr(r["g"]), nat.ptr(hv),
                     0 if hv is None else hv.shape[1], nat.ptr(Dmat), float(t), n, nat.ptr(eloc), nat.stream())
        return eloc[None, ...], r["J"], r["logp"][None, ...]
