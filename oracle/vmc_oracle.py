"""CPU oracle for the TDVP / stochastic-reconfiguration hot path of RehMoritz/vmc_pde.

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.
The product path (``vmc_pde_amd``) never imports it and has no CPU fallback.

It is a plain torch-fp64-on-CPU restatement of the reference algorithm *as written*
(``/root/reference/vmc_fluids``), with every derivative obtained by autodiff
(``torch.func``) exactly where the reference uses ``jax.grad`` / ``jacrev(jacfwd)``, so it
is independent of the hand-derived derivatives in the HIP kernels.

PARITY STATUS: **vector-level parity with the JAX reference is unpinned.**  The reference is
pure Python on jax==0.2.18 / flax==0.3.6 / mpi4py, none of which exist in the build image
(ordinary ModuleNotFoundError, no network), and it ships no tests, golden vectors or fixtures.
What pins this oracle instead (tests/test_oracle_known_answers.py):
  * the mathematical definition, read line by line from the cited reference sources,
  * analytic known answers KA-1..KA-4, KA-7 of SURVEY.md section 4,
  * the published HDF5 trajectories of the reference (statistical, tests/golden/published_*.npz).

Every function cites the reference file:line it follows.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
from torch.func import grad, jacfwd, jacrev, vmap

tReal = torch.float64  # global_defs.py:8 (tReal = np.float64), main.py:2 (jax_enable_x64)


# --------------------------------------------------------------------------------------------
# util.py:21-26
# --------------------------------------------------------------------------------------------
def build_cov_matrix(L_para: torch.Tensor, L_diag: torch.Tensor, dim: int) -> torch.Tensor:
    """util.py:21-26 -- strict upper triangle (row-major triu_indices(k=1)) + diag(exp(L_diag)); L L^T."""
    L = torch.zeros((dim, dim), dtype=tReal)
    iu = torch.triu_indices(dim, dim, offset=1)
    if L_para.numel() > 0:
        L = L.index_put((iu[0], iu[1]), L_para)
    L = L + torch.diag(torch.exp(L_diag))
    return L @ L.T


# --------------------------------------------------------------------------------------------
# Parameter pytree <-> flat vector  (var_state.py:25-27, 94-108; JAX pytree order = sorted dict keys)
# --------------------------------------------------------------------------------------------
@dataclass
class NetSpec:
    """Static description of an INNwProb (net.py:185-207) built by VarState.init_net (var_state.py:110-124)."""
    dim: int
    inds_up: List[List[int]]
    inds_down: List[List[int]]
    intmediate: Tuple[int, ...] = (3,)
    offset: Sequence[float] = ()
    latentSpaceName: str = "Gauss"
    # SingleBlock flags, net.py:69-72 (INN never forwards them, net.py:166: defaults are what runs)
    jac_eq_1: bool = False
    different_add: bool = False
    no_add: bool = True
    global_change: bool = False   # net.py:72: per-block affine map y -> global_scale * y + global_offset
    dense_dtype: str = "float64"  # flax 0.3.6 `nn.Dense` as called at net.py:53-61 has its own `dtype` (default float32) and casts
                                  # inputs, kernel and bias to it even under jax_enable_x64 (main.py:2).  "float32" restates that reading:
                                  # every conditioner MLP end to end in fp32, its output promoted when it meets the fp64 coordinates
                                  # (net.py:94-109).  "float64" (default) is the reading the product implements; which one the published
                                  # records support is tested in tests/test_oracle_published_run.py and recorded in DESIGN.md section 2.

    @property
    def depth(self) -> int:
        return len(self.inds_up)

    @property
    def n_dist(self) -> int:  # net.py:197-198
        return {"Gauss": 0, "Student_t": 1}[self.latentSpaceName]


def trafo_names(spec: NetSpec) -> List[str]:
    return ["s1", "s2", "t1", "t2"] if spec.different_add else ["s1", "s2"]  # net.py:75-79


def trafo_io(spec: NetSpec, b: int, name: str) -> Tuple[int, int]:
    """(in, out) widths of SingleTrafo `name` in block b: net.py:75-79, 89, 100."""
    nu, nd = len(spec.inds_up[b]), len(spec.inds_down[b])
    # s1/t1 act on v1 (|up|) and output |down|; s2/t2 act on u2 (|down|) and output |up|
    return (nu, nd) if name in ("s1", "t1") else (nd, nu)


def param_layout(spec: NetSpec) -> List[Tuple[str, Tuple[int, ...]]]:
    """Flat parameter order = jax.tree_flatten order = sorted dict keys at every level
    (var_state.py:25, 106-108).  Returns [(path, shape)]."""
    d = spec.dim
    out: List[Tuple[str, Tuple[int, ...]]] = []
    out.append(("L", ((d * d - d) // 2,)))          # net.py:201
    out.append(("L_diag", (d,)))                    # net.py:202
    out.append(("dist_params", (spec.n_dist,)))     # net.py:204
    out.append(("mu", (d,)))                        # net.py:203
    block_names = sorted(f"blocks_{b}" for b in range(spec.depth))  # string sort: blocks_10 < blocks_2
    for bn in block_names:
        b = int(bn.split("_")[1])
        if spec.global_change:                      # net.py:80-82; 'global_offset' < 'global_scale' < 's1' in the sorted dict
            out.append((f"myINN/{bn}/global_offset", (d,)))
            out.append((f"myINN/{bn}/global_scale", (1,)))
        for tn in sorted(trafo_names(spec)):
            n_in, n_out = trafo_io(spec, b, tn)
            widths = [n_in] + list(spec.intmediate) + [n_out]
            dense_names = sorted(f"Dense_{k}" for k in range(len(widths) - 1))
            for dn in dense_names:
                k = int(dn.split("_")[1])
                out.append((f"myINN/{bn}/{tn}/{dn}/bias", (widths[k + 1],)))
                out.append((f"myINN/{bn}/{tn}/{dn}/kernel", (widths[k], widths[k + 1])))  # (in,out) row-major
    return out


def num_parameters(spec: NetSpec) -> int:
    return sum(int(np.prod(s)) for _, s in param_layout(spec))


def unflatten(spec: NetSpec, theta: torch.Tensor) -> Dict[str, torch.Tensor]:
    """var_state.py:94-101 set_parameters."""
    p, start = {}, 0
    for name, shape in param_layout(spec):
        n = int(np.prod(shape))
        p[name] = theta[start:start + n].reshape(shape)
        start += n
    assert start == theta.numel()
    return p


def flatten(spec: NetSpec, p: Dict[str, torch.Tensor]) -> torch.Tensor:
    """var_state.py:103-108 get_parameters / flatten_tree."""
    return torch.cat([p[name].reshape(-1) for name, _ in param_layout(spec)])


def random_partitions(dim: int, depth: int, seed: int) -> Tuple[List[List[int]], List[List[int]]]:
    """var_state.py:114-119: ind_up = random d//2 subset (in random order, like jax.random.choice without
    replacement), ind_down = sorted complement.  The JAX threefry stream is not reproducible here (H4), so
    a numpy Generator stands in; parity tests always inject the partitions explicitly."""
    rng = np.random.default_rng(seed)
    ups, downs = [], []
    for _ in range(depth):
        up = [int(v) for v in rng.permutation(dim)[: dim // 2]]
        down = [i for i in range(dim) if i not in up]
        ups.append(up)
        downs.append(down)
    return ups, downs


def init_parameters(spec: NetSpec, seed: int, out_scale: float = 1e-5, hidden_scale: float = 1.0) -> torch.Tensor:
    """net.py:39-41, 48-49, 55-56, 60-61, 201-204: hidden kernels U(-1,1), output kernel U(-1e-5,1e-5),
    biases 0, L = L_diag = mu = dist_params = 0."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    n_dense = len(spec.intmediate) + 1
    for name, shape in param_layout(spec):
        if name.endswith("/kernel"):
            k = int(name.split("/")[-2].split("_")[1])
            scale = out_scale if k == n_dense - 1 else hidden_scale
            p[name] = 2 * scale * (torch.rand(shape, generator=g, dtype=tReal) - 0.5)
        elif name.endswith("/global_scale"):
            p[name] = torch.ones(shape, dtype=tReal)                       # net.py:81 initializers.ones
        else:
            p[name] = torch.zeros(shape, dtype=tReal)
    return flatten(spec, p)


# --------------------------------------------------------------------------------------------
# net.py
# --------------------------------------------------------------------------------------------
def gauss_logpdf(x, S, mu, dim):
    """net.py:14-20."""
    x = x - mu
    arg = x @ torch.linalg.inv(S) @ x
    return -0.5 * (dim * math.log(2 * math.pi) + torch.log(torch.linalg.det(S)) + arg)


def student_t_logpdf(x, S, mu, dist_params):
    """net.py:27-36 (note: no log det S term, as in the reference)."""
    p = x.shape[0]
    nu = torch.exp(dist_params[0]) + 1e0
    x = x - mu
    return (torch.lgamma((nu + p) / 2) - torch.lgamma(nu / 2) - p / 2 * torch.log(nu * math.pi)
            - (nu + p) / 2 * torch.log(1 + x @ torch.linalg.inv(S) @ x / nu))


def single_trafo(p: Dict[str, torch.Tensor], prefix: str, n_hidden: int, x: torch.Tensor, alpha: float = 10.0, dense_dtype: str = "float64"):
    """net.py:53-61: tanh MLP, output alpha*tanh(Dense).  dense_dtype="float32": every nn.Dense casts inputs, kernel and bias to fp32
    (NetSpec.dense_dtype); the casts are differentiable, so the autodiff derivatives are those of the fp32 computation."""
    cast = (lambda v: v.to(torch.float32)) if dense_dtype == "float32" else (lambda v: v)
    out_dtype = x.dtype
    for k in range(n_hidden):
        x = torch.tanh(cast(x) @ cast(p[f"{prefix}/Dense_{k}/kernel"]) + cast(p[f"{prefix}/Dense_{k}/bias"]))
    k = n_hidden
    return (alpha * torch.tanh(cast(x) @ cast(p[f"{prefix}/Dense_{k}/kernel"]) + cast(p[f"{prefix}/Dense_{k}/bias"]))).to(out_dtype)


def single_block(spec: NetSpec, p, b: int, x: torch.Tensor, inv: bool):
    """net.py:84-153, including the global_change option (net.py:72,80-82,115-116,150-151) exactly as written: the forward
    direction returns scale * result + offset, and the inverse direction returns (result - offset) / scale AFTER the inverse
    coupling steps -- which is not the inverse map of the forward direction unless scale = 1, offset = 0 (the affine step would
    have to be undone BEFORE them).  INN never enables the option (net.py:166), so the reference never met this."""
    up, dn = spec.inds_up[b], spec.inds_down[b]
    nh = len(spec.intmediate)
    pre = f"myINN/blocks_{b}"
    s1 = lambda v: single_trafo(p, f"{pre}/s1", nh, v, dense_dtype=spec.dense_dtype)
    s2 = lambda v: single_trafo(p, f"{pre}/s2", nh, v, dense_dtype=spec.dense_dtype)
    t1 = lambda v: single_trafo(p, f"{pre}/t1", nh, v, dense_dtype=spec.dense_dtype)
    t2 = lambda v: single_trafo(p, f"{pre}/t2", nh, v, dense_dtype=spec.dense_dtype)
    if not inv:
        u1, u2 = x[up], x[dn]
        s2_u2 = s2(u2)
        if spec.jac_eq_1:
            v1 = u1 + s2_u2
            s2_u2 = torch.zeros_like(s2_u2)
        elif spec.different_add:
            v1 = u1 * torch.exp(s2_u2) + t2(u2)
        elif spec.no_add:
            v1 = u1 * torch.exp(s2_u2)
        else:
            v1 = u1 * torch.exp(s2_u2) + s2_u2
        s1_v1 = s1(v1)
        if spec.jac_eq_1:
            v2 = u2 + s1_v1
            s1_v1 = torch.zeros_like(s1_v1)
        elif spec.different_add:
            v2 = u2 * torch.exp(s1_v1) + t1(v1)
        elif spec.no_add:
            v2 = u2 * torch.exp(s1_v1)
        else:
            v2 = u2 * torch.exp(s1_v1) + s1_v1
        res = torch.zeros_like(x)
        res = res.index_put((torch.tensor(up),), v1)
        res = res.index_put((torch.tensor(dn),), v2)
        if spec.global_change:                                           # net.py:115-116
            gs, go = p[f"{pre}/global_scale"], p[f"{pre}/global_offset"]
            return gs * res + go, (s2_u2.sum() + s1_v1.sum()) + torch.log(gs[0]) * (len(up) + len(dn))
        return res, s2_u2.sum() + s1_v1.sum()
    else:
        v1, v2 = x[up], x[dn]
        s1_v1 = s1(v1)
        if spec.jac_eq_1:
            u2 = v2 - s1_v1
            s1_v1 = torch.zeros_like(s1_v1)
        elif spec.different_add:
            u2 = (v2 - t1(v1)) * torch.exp(-s1_v1)
        elif spec.no_add:
            u2 = v2 * torch.exp(-s1_v1)
        else:
            u2 = (v2 - s1_v1) * torch.exp(-s1_v1)
        s2_u2 = s2(u2)
        if spec.jac_eq_1:
            u1 = v1 - s2_u2
            s2_u2 = torch.zeros_like(s2_u2)
        elif spec.different_add:
            u1 = (v1 - t2(u2)) * torch.exp(-s2_u2)
        elif spec.no_add:
            u1 = v1 * torch.exp(-s2_u2)
        else:
            u1 = (v1 - s2_u2) * torch.exp(-s2_u2)
        res = torch.zeros_like(x)
        res = res.index_put((torch.tensor(up),), u1)
        res = res.index_put((torch.tensor(dn),), u2)
        if spec.global_change:                                           # net.py:150-151
            gs, go = p[f"{pre}/global_scale"], p[f"{pre}/global_offset"]
            return (res - go) / gs, -(s1_v1.sum() + s2_u2.sum()) - torch.log(gs[0]) * (len(up) + len(dn))
        return res, -(s1_v1.sum() + s2_u2.sum())


def inn(spec: NetSpec, p, x: torch.Tensor, inv: bool):
    """net.py:168-182."""
    log_jac = torch.zeros((), dtype=tReal)
    order = range(spec.depth) if not inv else range(spec.depth - 1, -1, -1)
    for b in order:
        x, lj = single_block(spec, p, b, x, inv)
        log_jac = log_jac + lj
    return x, log_jac


def latent_logpdf(spec: NetSpec, p, z: torch.Tensor):
    """net.py:212 / 215: latent_space_dist(z - offset, {S, mu, dist_params})."""
    offset = torch.as_tensor(np.asarray(spec.offset, dtype=np.float64))
    S = build_cov_matrix(p["L"], p["L_diag"], spec.dim)
    if spec.latentSpaceName == "Gauss":
        return gauss_logpdf(z - offset, S, p["mu"], spec.dim)
    return student_t_logpdf(z - offset, S, p["mu"], p["dist_params"])


def inn_w_prob(spec: NetSpec, theta: torch.Tensor, x: torch.Tensor, evaluate: bool = True, inv: bool = False):
    """net.py:209-217 INNwProb.__call__ on ONE sample x (d,)."""
    p = unflatten(spec, theta)
    if evaluate:
        z, log_jac = inn(spec, p, x, inv)
        return latent_logpdf(spec, p, z) + log_jac
    p_latent_log = latent_logpdf(spec, p, x)
    y, log_jac = inn(spec, p, x, inv)
    return y, p_latent_log - log_jac


# --------------------------------------------------------------------------------------------
# var_state.py
# --------------------------------------------------------------------------------------------
class VarStateOracle:
    """var_state.py:18-124, batched over samples with vmap exactly as J1..J6.
    Arrays carry the reference's leading device axis of size 1."""

    def __init__(self, spec: NetSpec, theta: torch.Tensor):
        self.spec = spec
        self.dim = spec.dim
        self.theta = theta.clone().to(tReal)
        self.numParameters = num_parameters(spec)
        assert self.theta.numel() == self.numParameters
        f = lambda x, th: inn_w_prob(spec, th, x)
        self._eval = vmap(f, in_dims=(0, None))                                            # var_state.py:29
        self._grads_coords = vmap(grad(f, argnums=(0, 1)), in_dims=(0, None))               # var_state.py:31
        self._hessian = vmap(jacrev(jacfwd(f, argnums=0), argnums=0), in_dims=(0, None))    # var_state.py:32
        self._latent = vmap(lambda z, th: inn_w_prob(spec, th, z, evaluate=False, inv=True), in_dims=(0, None))  # :33

    def set_parameters(self, theta):
        self.theta = theta.clone().to(tReal)

    def get_parameters(self):
        return self.theta.clone()

    def __call__(self, coords: torch.Tensor, mode: str = "eval"):
        x = coords[0]
        if mode == "eval":
            return self._eval(x, self.theta)[None]
        if mode == "eval_coordgrads":   # var_state.py:55-64
            val = self._eval(x, self.theta)
            gx, gth = self._grads_coords(x, self.theta)
            return val[None], gx[None], gth[None]
        raise ValueError(mode)

    def hessian(self, coords):          # var_state.py:66-67
        return self._hessian(coords[0], self.theta)[None]

    def flow_sample(self, z):           # var_state.py:79 (latent z already offset-shifted, sampler.py:86)
        x, logp = self._latent(z[0], self.theta)
        return x[None], logp[None]

    def dist_params(self):              # var_state.py:77
        p = unflatten(self.spec, self.theta)
        return {"S": build_cov_matrix(p["L"], p["L_diag"], self.dim), "mu": p["mu"], "dist_params": p["dist_params"]}


# --------------------------------------------------------------------------------------------
# sampler.py
# --------------------------------------------------------------------------------------------
def latent_from_normals(spec: NetSpec, theta: torch.Tensor, normals: torch.Tensor, chi2: torch.Tensor | None = None):
    """sampler.py:25-34, 86 with the random draws injected (H4): normals (N,d) iid N(0,1), chi2 (N,) ~ chi^2_nu.
    multivariate_normal(mu, S) is realised as mu + chol_lower(S) @ n (any factor with F F^T = S gives the same law;
    the product uses the same lower Cholesky factor so vectors agree)."""
    p = unflatten(spec, theta)
    S = build_cov_matrix(p["L"], p["L_diag"], spec.dim)
    C = torch.linalg.cholesky(S)
    offset = torch.as_tensor(np.asarray(spec.offset, dtype=np.float64))
    y = normals @ C.T
    if spec.latentSpaceName == "Gauss":
        z = p["mu"] + y
    else:
        nu = torch.exp(p["dist_params"][0]) + 1.0
        z = torch.sqrt(nu / chi2)[:, None] * y + p["mu"]
    return (z + offset)[None]


def cos_dist(x: np.ndarray, offset: np.ndarray) -> np.ndarray:
    """sampler.py:20-22 (vectorised over chains): log(0.5*(1+cos(pi*min(1, 4|x-offset|))))."""
    r = np.minimum(1.0, 4.0 * np.sqrt(np.sum((x - offset) ** 2, axis=-1)))
    with np.errstate(divide="ignore"):
        return np.log(0.5 * (1.0 + np.cos(np.pi * r)))


def radial_update_prop(u_r: np.ndarray, u_phi: np.ndarray, bound: float, offset: np.ndarray) -> np.ndarray:
    """sampler.py:37-45 with the two uniform draws injected."""
    r = np.sqrt(u_r) * bound
    phi = 2 * np.pi * u_phi
    return np.stack([r * np.cos(phi), r * np.sin(phi)], axis=-1) + offset


def mcmc_chain(states0: np.ndarray, u_r: np.ndarray, u_phi: np.ndarray, u_acc: np.ndarray, bound: float, offset: np.ndarray):
    """sampler.py:88-111 independence-Metropolis scan; uniforms injected: u_* have shape (steps, chains).
    bernoulli(key, P) is realised as u < P.  Returns (states (steps*chains, 2), numAccepted)."""
    states = states0.copy()
    out = np.empty((u_r.shape[0],) + states.shape)
    n_acc = 0
    for s in range(u_r.shape[0]):
        new = radial_update_prop(u_r[s], u_phi[s], bound, offset)
        with np.errstate(invalid="ignore", over="ignore"):
            P = np.exp(cos_dist(new, offset) - cos_dist(states, offset))
        acc = u_acc[s] < P
        n_acc += int(acc.sum())
        states = np.where(acc[:, None], new, states)
        out[s] = states
    return out.reshape(-1, 2), n_acc


# --------------------------------------------------------------------------------------------
# evolutionEq.py
# --------------------------------------------------------------------------------------------
def velocity_field_MLPaper(params, coord, t):
    """evolutionEq.py:23-27."""
    x, y = coord[0], coord[1]
    c = math.cos(math.pi * t / params["T"])
    return torch.stack([-torch.sin(math.pi * x) ** 2 * torch.sin(2 * math.pi * y) * c,
                        torch.sin(math.pi * y) ** 2 * torch.sin(2 * math.pi * x) * c])


def velocity_field_hamiltonian(params, coord, t):
    """evolutionEq.py:30-45.  The reference evaluates jax.grad(H) with the default coupled=False; the ring-coupled branch
    (:33-38) is restated too and taken when params carries "coupled": True."""
    def H(x):
        if params.get("coupled", False):
            xs, ps = x[0::2], x[1::2]
            return (params["m"] * params["omega"] ** 2 / 2 * torch.sum((xs - torch.roll(xs, 1)) ** 2)
                    + torch.sum(ps ** 2) / (2.0 * params["m"]) + params["lam"] * torch.sum(xs ** 4))
        return (params["m"] * params["omega"] ** 2 / 2 * torch.sum(x[0::2] ** 2)
                + torch.sum(x[1::2] ** 2) / (2.0 * params["m"]) + params["lam"] * torch.sum(x[0::2] ** 4))
    grads = grad(H)(coord)
    n = coord.shape[0] // 2
    mat = torch.kron(torch.eye(n, dtype=tReal), torch.tensor([[0.0, 1.0], [-1.0, 0.0]], dtype=tReal))
    return mat @ grads


def random_D_matrix(dim: int, seed: int = 0) -> torch.Tensor:
    """evolutionEq.py:18-20: A^T A with A ~ N(0,1) (JAX key 0 is not reproducible; seeded torch stands in)."""
    g = torch.Generator().manual_seed(seed)
    A = torch.randn((dim, dim), generator=g, dtype=tReal)
    return A.T @ A


class EvolutionEquationOracle:
    """evolutionEq.py:48-119."""

    def __init__(self, dim: int = 2, name: str = "diffusion"):
        self.dim, self.name = dim, name
        self.eqParams = {  # evolutionEq.py:61-77
            "diffusion": {"D": 1},
            "diffusion_anisotropic": {"D": random_D_matrix(dim)},
            "diffusion_drift": {"D": 1, "mu": 4},
            "advection_paper": {"params": {"T": 5}, "vel_field": velocity_field_MLPaper},
            "advection_hamiltonian": {"params": {"m": 1.0, "omega": 1.0, "lam": 0.0}, "vel_field": velocity_field_hamiltonian},
            "advection_hamiltonian_wDiss": {"params": {"m": 1.0, "omega": 1.0, "T": 10.0, "gamma": 1.0, "lam": 0.0},
                                            "vel_field": velocity_field_hamiltonian},
        }

    def _adv(self, coord_grads, configs, t):  # evolutionEq.py:79
        e = self.eqParams[self.name]
        f = lambda g, x: -g @ e["vel_field"](e["params"], x, t)
        return vmap(f)(coord_grads[0], configs[0])[None]

    def __call__(self, vState: VarStateOracle, configs: torch.Tensor, t: float):
        logProbs, coord_grads, param_grads = vState(configs, mode="eval_coordgrads")
        e = self.eqParams[self.name]
        if self.name == "diffusion":            # :84-87
            tg = e["D"] * ((coord_grads ** 2).sum(-1) + torch.einsum("abii->ab", vState.hessian(configs)))
        elif self.name == "diffusion_drift":    # :89-94
            tg = (e["D"] * ((coord_grads ** 2).sum(-1) + torch.einsum("abii->ab", vState.hessian(configs)))
                  + e["mu"] * coord_grads.sum(-1))
        elif self.name == "diffusion_anisotropic":  # :96-100
            tg = (torch.einsum("abi,ij,abj->ab", coord_grads, e["D"], coord_grads)
                  + torch.einsum("abij,ji->ab", vState.hessian(configs), e["D"]))
        elif self.name in ("advection_paper", "advection_hamiltonian"):  # :102-105
            tg = self._adv(coord_grads, configs, t)
        elif self.name == "advection_hamiltonian_wDiss":  # :107-119
            pr = e["params"]
            H = vState.hessian(configs)
            adv = self._adv(coord_grads, configs, t)
            diff = pr["m"] * pr["gamma"] * pr["T"] * ((coord_grads[:, :, 1::2] ** 2).sum(-1)
                                                      + torch.einsum("abii->ab", H[:, :, 1::2, 1::2]))
            damp = pr["gamma"] * (configs[:, :, 1::2] * coord_grads[:, :, 1::2]).sum(-1)
            tg = adv + diff + damp
        else:
            raise KeyError(self.name)
        return tg, param_grads, logProbs


# --------------------------------------------------------------------------------------------
# mpi_wrapper.py (single process: Allreduce is the identity) + tdvp.py
# --------------------------------------------------------------------------------------------
def distribute_sampling(numSamples, commSize=1, rank=0, localDevices=None, numChainsPerDevice=1):
    """mpi_wrapper.py:68-110.  Returns (samples per device/chain for this rank, globNumSamples)."""
    spp = numSamples // commSize
    if rank < numSamples % commSize:
        spp += 1
    if localDevices is None:
        return spp, numSamples
    ncpp = localDevices * numChainsPerDevice
    spc = lambda s: (s + ncpp - 1) // ncpp
    a = numSamples % commSize
    glob = (a * spc(1 + numSamples // commSize) + (commSize - a) * spc(numSamples // commSize)) * ncpp
    return spc(spp), glob


@dataclass
class TDVPOracle:
    """tdvp.py:20-94 algebra on already-evaluated (Eloc, gradients, logProbs) -- the reference as written:
    centred copies, two un-symmetrised Gram GEMMs, host eigh, EO@V GEMM, pseudo-inverse."""
    useSNR: bool = False
    snrTol: float = 2e0
    svdTol: float = 1e-11
    diagonalShift: float = 0.0
    frozen: tuple = ()       # NOT in the reference's tdvp.py: flat indices whose Jacobian columns are identically zero.  Emulates the
                             # latent covariance S = I + A A^T the published diffusion runs were made with (main.py:55-57 "latent
                             # space covariance matrix: Diffusion: np.eye(..) + A @ A.T"): at the initial A = 0, dS = dA A^T + A dA^T
                             # = 0, so dlogp/dA = 0, the pseudo-inverse (tdvp.py:82) never moves A, and the run equals the
                             # checked-out L L^T model (util.py:21-26) with L, L_diag held at 0.

    def get_tdvp_equation(self, Eloc, gradients, logProbs, N):
        if len(self.frozen):
            gradients = gradients.clone()
            gradients[..., list(self.frozen)] = 0.0
        self.ElocMean = Eloc.sum() / N                                   # :37  mpi_wrapper.py:166-193
        self.ElocMeanAbs = Eloc.abs().sum() / N                          # :38
        self.ElocVar = ((Eloc - self.ElocMean) ** 2).sum() / N           # :39  mpi_wrapper.py:196-245
        Eloc = Eloc - self.ElocMean                                      # :40
        gradientsMean = gradients.sum(dim=(0, 1)) / N                    # :41
        gradients = gradients - gradientsMean                            # :42
        EOdata = Eloc[..., None] * gradients                             # :44
        self.F0 = EOdata.sum(dim=(0, 1)) / N                             # :45
        O = gradients[0]
        self.S0 = (O.T @ O) / N                                          # :46  mpi_wrapper.py:21-25,248-274
        Ow = logProbs[0][:, None] * O                                    # :33
        self.SExp = (Ow.T @ Ow) / N                                      # :47
        S, F = self.S0, self.F0
        if self.diagonalShift > 1e-10:                                   # :50-51
            S = S + torch.diag(self.diagonalShift * torch.diag(S))
        return S, F, EOdata

    def transform_to_eigenbasis(self, S, F, EOdata, N):
        ev, V = np.linalg.eigh(S.numpy())                                # :61-64 host LAPACK
        self.ev, self.V = torch.from_numpy(ev), torch.from_numpy(V)
        self.VtF = self.V.T @ F                                          # :66
        EO = EOdata[0] @ self.V                                          # :68
        m = EO.sum(0) / N
        self.rhoVar = ((EO - m) ** 2).sum(0) / N                         # :70
        self.snr = torch.sqrt(torch.abs(N * (self.VtF * self.VtF) / self.rhoVar))  # :71

    def solve(self, Eloc, gradients, logProbs, N=None):
        N = Eloc.numel() if N is None else N
        self.S, F, Fdata = self.get_tdvp_equation(Eloc, gradients, logProbs, N)   # :75
        self.transform_to_eigenbasis(self.S, F, Fdata, N)                          # :79
        rel = torch.abs(self.ev / self.ev[-1])
        self.invEv = torch.where(rel > 1e-14, 1.0 / self.ev, torch.zeros_like(self.ev))   # :82
        regularizer = 1.0 / (1.0 + (self.svdTol / rel) ** 6)                      # :85
        if self.useSNR:
            regularizer = regularizer * (1.0 / (1.0 + (self.snrTol / self.snr) ** 6))  # :87-89
        update = self.V @ (self.invEv * regularizer * self.VtF)                   # :91
        tdvp_error = 1 + (update @ self.S0 @ update - 2 * self.F0 @ update) / torch.mean(Eloc ** 2)  # :93
        resid = torch.linalg.norm(self.S @ update - F) / torch.linalg.norm(F)     # :94
        self.solverResidual, self.tdvp_error = resid, tdvp_error
        return update, resid, tdvp_error

    def rhs(self, theta, t, psi: VarStateOracle, evolutionEq: EvolutionEquationOracle, z: torch.Tensor):
        """tdvp.py:96-128 with the latent draw z (1,N,d) injected instead of psi.sample's RNG."""
        psi.set_parameters(theta)                                        # :101
        x, logp_s = psi.flow_sample(z)                                   # :117 (var_state.py:79)
        Eloc, J, logp = evolutionEq(psi, x, t)                           # :122
        update, resid, err = self.solve(Eloc, J, logp)                   # :127
        return update, dict(x=x, logp_sample=logp_s, Eloc=Eloc, J=J, logp=logp)

    def observables(self, sampleConfigs, logProbs, Eloc):
        """tdvp.py:143-150."""
        info = {}
        mean = sampleConfigs.mean(dim=(0, 1), keepdim=True)
        info["x1"] = mean[0, 0]
        X = sampleConfigs[0]
        info["covar"] = (X - X.mean(0)).T @ (X - X.mean(0)) / X.shape[0]
        info["entropy"] = -logProbs.mean()
        for m in [3, 4, 5, 6]:
            info[f"x{m}"] = ((sampleConfigs - mean) ** m).mean(dim=(0, 1))
        info["max_grad"] = Eloc.max()
        return info

    @staticmethod
    def ball_points(normals, uniforms):
        """tdvp.py:154-155 with the draws injected: directions normals / |normals| (1,n,d), radii uniforms^(1/d) (1,n) -- uniform
        points in the unit ball.  (The reference draws both arrays from the SAME key `psi.sampler.key`; threefry streams are not
        reproducible here (H4), so the two arrays are inputs.)"""
        dim = normals.shape[-1]
        return normals / torch.linalg.norm(normals, dim=-1, keepdim=True) * uniforms[..., None] ** (1 / dim)

    @staticmethod
    def small_ball_integrals(psi, samples):
        """tdvp.py:157-162: Monte-Carlo integral of exp(logp) over balls of radius lim * sqrt(T), T = 10, lim in (1, 0.5, 0.1),
        from uniform unit-ball points `samples` (1,n,d): mean(p(lim * samples)) * volume of the ball."""
        info = {}
        dim = samples.shape[-1]
        for lim in [1, 0.5, 0.1]:
            lim_normal = lim
            T = 10
            lim = lim * math.sqrt(T)                                                             # :160
            sphere_volume = math.pi ** (dim / 2) / math.gamma(dim / 2 + 1) * lim ** dim          # :161
            info[f"integral_{lim_normal}sigma"] = torch.mean(torch.exp(psi(lim * samples))) * sphere_volume   # :162
        return info


# --------------------------------------------------------------------------------------------
# stepper.py (value semantics: every `y +=` of the reference rebinds an immutable jnp array)
# --------------------------------------------------------------------------------------------
def heun_step(f, y, t, dt):
    """stepper.py:133-139 with dt already updated (:131)."""
    k0, _ = f(y, t, 0)
    k1, info = f(y + dt * k0, t + dt, 1)
    return y + 0.5 * dt * (k0 + k1), info


def euler_step(f, y, t, dt):
    """stepper.py:141-145."""
    k0, info = f(y, t, 0)
    return y + dt * k0, info


def fixed_stepper_dt(dt, increase_fac, maxStep):
    """stepper.py:131."""
    return min(dt * increase_fac, maxStep)


def adaptive_heun_step(f, y, t, dt, tol, maxStep, normFunction, get_SExp):
    """stepper.py:45-91.  f(y,t,intStep)->(k,info); get_SExp() returns f.SExp after the last call."""
    fe = 0.5
    yInitial = y.clone()
    while fe < 1.0:
        y = yInitial.clone()
        k0, info = f(y, t, 0)
        y = y + dt * k0
        k1, _ = f(y, t + dt, 1)
        dy0 = 0.5 * dt * (k0 + k1)
        y = y - 0.5 * dt * k0
        k10, _ = f(y, t + 0.5 * dt, 2)
        dy1 = 0.25 * dt * (k0 + k10)
        y = yInitial + dy1
        k01, _ = f(y, t + 0.5 * dt, 3)
        y = y + 0.5 * dt * k01
        k11, _ = f(y, t + dt, 4)
        dy1 = dy1 + 0.25 * dt * (k01 + k11)
        updateDiff = normFunction(dy1 - dy0, get_SExp())
        fe = float(tol / updateDiff)
        tmp = max(0.2, 0.9 * fe ** 0.33333)
        tmp = min(tmp, 2.0)
        realDt = dt
        dt = min(dt * tmp, maxStep)
    return yInitial + dy1, realDt, dt, info


def norm_fun(v, S):
    """main.py:24-26."""
    return v @ S @ v
