"""Variational state (reference: vmc_fluids/var_state.py).

Keeps the reference's call surface -- `VarState(sampler, dim, initKey, depth, network_args=...)`, `__call__(coords,
mode)`, `hessian`, `sample`, `integrate`, `set_parameters` / `get_parameters`, attributes `net, params, paramShapes,
numParameters` -- on top of the HIP kernels of csrc/flow.hip.  Arrays keep the reference's leading "device" axis
of size 1; tensors are fp64 on the GPU.

Memory layout: the per-sample parameter Jacobian is produced PARAMETER-MAJOR in HBM (JT[p][n], row pitch `ldj`,
rows padded with zeros to a multiple of 128 for the MFMA Gram kernel).  `__call__(..., "eval_coordgrads")` returns
it as the reference's (1, N, P) array, which is a strided view of that buffer (no copy).  The buffers are cached
per sample count and overwritten by the next evaluation with the same N.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _native as nat
from . import global_defs
from . import net
from . import util


def _round_up(v, m):
    return (v + m - 1) // m * m


class _LatentParams(dict):
    """The `dist_params` dict of var_state.py:77 ({"S", "mu", "dist_params"}).  The covariance S = L L^T (util.py:21-26) is only
    built when a sampler actually reads it: the native sampler rebuilds the factor on the device from the flat parameter vector
    (csrc/flow.hip::latent_prepare_kernel) and never does -- ten tiny launches per TDVP call otherwise."""

    def __init__(self, owner, dim, model, theta):
        super().__init__(_model=model, _theta=theta)
        self._owner, self._dim = owner, dim

    def __missing__(self, key):
        p = self._owner.params["params"]          # (builds the parameter tree: only the generic sampler path comes here)
        if key == "S":
            self["S"] = util.build_cov_matrix(p["L"], p["L_diag"], self._dim)
            return self["S"]
        if key in ("mu", "dist_params"):
            self[key] = p[key]
            return self[key]
        raise KeyError(key)

    def __contains__(self, key):
        return key in ("S", "mu", "dist_params") or super().__contains__(key)


class VarState:

    def __init__(self, sampler, dim, *args, network_args={}, inds_up=None, inds_down=None, jacobian_dtype=torch.float64):
        self.sampler = sampler
        self.dim = dim
        self.device = global_defs.my_device()
        assert jacobian_dtype in (torch.float64, torch.float32)
        self.jacobian_dtype = jacobian_dtype      # float32 = BASELINE config 5 (fp32 Jacobian, fp64 S/F accumulation)
        # True: the forward pass of flow_eval keeps every network's activations in an HBM workspace and the reverse sweep reads them
        # back instead of re-evaluating the networks.  Identical results; measured SLOWER (C2 0.78 ms against 0.71 ms, C3 2.40 against
        # 2.32): the whole activation function costs 0.05 ms of the kernel's 0.71 (tools/diag/flow_where.sh), the kernel is bound by
        # its ~170 workgroup barriers per sample group, and the global round trip adds latency to that chain.  Off by default.
        self.keep_activations = False
        self.net, self._theta = self.init_net(network_args, *args, inds_up=inds_up, inds_down=inds_down)
        self.numParameters = self.net.numParameters                       # var_state.py:27
        self.paramShapes = [(int(np.prod(s)), tuple(s)) for _, _, s in self.net.layout]   # var_state.py:25
        self.netTreeDef = ["/".join(p) for p, _, _ in self.net.layout]    # var_state.py:26: the leaf order of the parameter tree
        self._params = None                                               # the nested dict of views, built on first access
        self._ws = {}
        self._dirs_cache = {}

    # ---- workspaces -------------------------------------------------------------------------------------------
    def _workspace(self, n):
        ws = self._ws.get(n)
        if ws is None:
            P = self.numParameters
            ldj = _round_up(max(n, 1), 16)
            ws = dict(ldj=ldj,
                      JT=torch.zeros((_round_up(P, nat.GRAM_TILE), ldj), dtype=self.jacobian_dtype, device=self.device),
                      logp=torch.empty(n, dtype=global_defs.tReal, device=self.device),
                      g=torch.empty((n, self.dim), dtype=global_defs.tReal, device=self.device))
            ws["gen"] = 0
            if len(self._ws) >= 2:   # at most two sample counts alive (TDVP + observables)
                self._ws.pop(next(iter(self._ws)))
            self._ws[n] = ws
        return ws

    def workspace_generation(self, n):
        """(identity, write count) of the per-N workspace: changes whenever `evaluate` overwrites logp / g / the Jacobian of that
        sample count, or the buffer is dropped.  TDVP's lazy SExp / rhoVar / snr check it before re-reading the Jacobian."""
        ws = self._ws.get(n)
        return None if ws is None else (id(ws), ws["gen"])

    def _dirs(self, dirs):
        """Device copy of a (D, d) direction set (cached by content)."""
        arr = np.ascontiguousarray(np.asarray(dirs, dtype=np.float64).reshape(-1, self.dim))
        key = arr.tobytes()
        t = self._dirs_cache.get(key)
        if t is None:
            t = torch.from_numpy(arr).to(self.device)
            self._dirs_cache[key] = t
        return t

    def evaluate(self, coords, dirs=None, want_g=True, want_J=True):
        """One fused pass of csrc/flow.hip over a batch: log p, grad_x, v_k^T H v_k for the given directions, and the
        parameter-major Jacobian.  Returns a dict of device tensors (J as the (1,N,P) strided view)."""
        x = torch.as_tensor(coords, dtype=global_defs.tReal, device=self.device).reshape(-1, self.dim).contiguous()
        n = x.shape[0]
        ws = self._workspace(n)
        ws["gen"] += 1
        D = 0
        hv = dirs_t = None
        if dirs is not None and len(dirs):
            dirs_t = self._dirs(dirs)
            D = dirs_t.shape[0]
            hv = torch.empty((n, D), dtype=global_defs.tReal, device=self.device)
        act = None
        if (want_g or want_J) and self.keep_activations:
            # activation workspace of the reverse sweep (hidden activations + outputs of every MLP, [unit][sample]): the forward
            # pass fills it, the reverse sweep reads it back instead of re-evaluating the networks
            if ws.get("act") is None:
                need = nat.lib().vmcpde_flow_eval_workspace_bytes(self.net.handle, n)
                ws["act"] = torch.empty(max(need // 8, 1), dtype=global_defs.tReal, device=self.device)
            act = ws["act"]
        with torch.cuda.device(self.device):
            nat.call("vmcpde_flow_eval", self.net.handle, nat.ptr(self._theta), nat.ptr(x), n, nat.ptr(dirs_t), D,
                     nat.ptr(ws["logp"]), nat.ptr(ws["g"]) if want_g else None, nat.ptr(hv),
                     nat.ptr(ws["JT"]) if want_J else None, ws["ldj"], 1 if self.jacobian_dtype == torch.float32 else 0,
                     nat.ptr(act), 0 if act is None else act.numel() * 8, nat.stream())
        P = self.numParameters
        return dict(x=x, logp=ws["logp"], g=ws["g"], hv=hv, JT=ws["JT"], ldj=ws["ldj"], n=n,
                    J=ws["JT"][:P, :n].T[None, ...] if want_J else None)

    def jt_buffer_of(self, gradients):
        """If `gradients` (1,N,P) is the strided view handed out by `evaluate`, return (JT, ldj) without copying."""
        n = gradients.shape[1]
        ws = self._ws.get(n)
        if ws is not None and gradients.data_ptr() == ws["JT"].data_ptr() and gradients.stride(1) == 1 \
                and gradients.stride(2) == ws["ldj"]:
            return ws["JT"], ws["ldj"]
        return None

    # ---- var_state.py:36-64 -----------------------------------------------------------------------------------
    def __call__(self, coords, mode="eval", avg=False):
        if mode == "eval":
            x = torch.as_tensor(coords, dtype=global_defs.tReal, device=self.device)
            lead = x.shape[:-1]
            value = self.net.apply(self._theta, x.reshape(-1, self.dim)).reshape(lead)
            return value.mean() if avg else value
        if mode == "eval_coordgrads":
            r = self.evaluate(coords)
            return r["logp"][None, ...], r["g"][None, ...], r["J"]
        if mode == "costfun":
            # var_state.py:45-53: negative log-likelihood and its parameter gradient (training only)
            r = self.evaluate(coords, want_g=False)
            value, grad = -r["logp"][None, ...], -r["J"]
            if avg:
                return value.mean(), grad.mean(dim=(0, 1))
            return value, grad
        raise ValueError(f"unknown mode {mode}")

    def hessian(self, coords):
        """var_state.py:66-67: full Hessian (1,N,d,d) from d(d+1)/2 directional second derivatives
        (H_ij = ((e_i+e_j)^T H (e_i+e_j) - H_ii - H_jj)/2)."""
        d = self.dim
        dirs, pairs = [np.eye(d)[i] for i in range(d)], []
        for i in range(d):
            for j in range(i + 1, d):
                v = np.zeros(d)
                v[i] = v[j] = 1.0
                dirs.append(v)
                pairs.append((i, j))
        r = self.evaluate(coords, dirs=np.stack(dirs), want_g=False, want_J=False)
        hv, n = r["hv"], r["n"]
        H = torch.zeros((n, d, d), dtype=global_defs.tReal, device=self.device)
        idx = torch.arange(d, device=self.device)
        H[:, idx, idx] = hv[:, :d]
        for k, (i, j) in enumerate(pairs):
            off = 0.5 * (hv[:, d + k] - hv[:, i] - hv[:, j])
            H[:, i, j] = off
            H[:, j, i] = off
        return H[None, ...]

    def real_space_prob(self, x, params=None):    # var_state.py:73-74
        return self.net.apply(self._theta if params is None else params, x)

    def sample(self, numSamples):
        """var_state.py:76-79: latent samples -> inverse flow -> (x (1,n,d), logp (1,n))."""
        dist_params = _LatentParams(self, self.dim, self.net, self._theta)
        z = self.sampler(numSamples, dist_params)
        return self.flow_sample(z)

    def draw_latent(self, numSamples):
        """The first half of `sample` (var_state.py:77-78): the sampler's latent points (1, n, d), shifted by the offset."""
        dist_params = _LatentParams(self, self.dim, self.net, self._theta)
        return self.sampler(numSamples, dist_params)

    def evaluate_from_latent(self, latent_space_samples, dirs=None):
        """`flow_sample` + `evaluate` in ONE launch for a batch given by its latent points (`vmcpde_flow_eval_from_latent`): the reverse sweep of
        csrc/flow.hip walks the flow backwards from the latent end with the exact inverse of every coupling half-step -- that IS the inverse flow
        z -> x of var_state.py:79 -- so it starts at z, produces x on its way, and the Taylor pass then runs forward from that x.  Returns
        `evaluate`'s dict; `x` is a fresh tensor the caller may keep.  Same numbers as the two launches to round-off (not bit for bit)."""
        z = torch.as_tensor(latent_space_samples, dtype=global_defs.tReal, device=self.device).reshape(-1, self.dim).contiguous()
        n = z.shape[0]
        ws = self._workspace(n)
        ws["gen"] += 1
        D = 0
        hv = dirs_t = None
        if dirs is not None and len(dirs):
            dirs_t = self._dirs(dirs)
            D = dirs_t.shape[0]
            hv = torch.empty((n, D), dtype=global_defs.tReal, device=self.device)
        x = torch.empty((n, self.dim), dtype=global_defs.tReal, device=self.device)
        with torch.cuda.device(self.device):
            nat.call("vmcpde_flow_eval_from_latent", self.net.handle, nat.ptr(self._theta), nat.ptr(z), n, nat.ptr(dirs_t), D, nat.ptr(x),
                     nat.ptr(ws["logp"]), nat.ptr(ws["g"]), nat.ptr(hv), nat.ptr(ws["JT"]), ws["ldj"],
                     1 if self.jacobian_dtype == torch.float32 else 0, nat.stream())
        P = self.numParameters
        return dict(x=x, logp=ws["logp"], g=ws["g"], hv=hv, JT=ws["JT"], ldj=ws["ldj"], n=n, J=ws["JT"][:P, :n].T[None, ...])

    def flow_sample(self, latent_space_samples):
        """var_state.py:33,79 (J5) for injected latent samples (1,n,d)."""
        z = torch.as_tensor(latent_space_samples, dtype=global_defs.tReal, device=self.device).reshape(-1, self.dim).contiguous()
        x, logp = self.net.apply(self._theta, z, evaluate=False, inv=True)
        return x[None, ...], logp[None, ...]

    def average_tree(self, tree, axis=(0, 1)):    # var_state.py:81-86: leaf-wise mean of a nested dict
        if isinstance(tree, dict):
            return {k: self.average_tree(v, axis) for k, v in tree.items()}
        return tree.mean(dim=axis)

    def integrate(self, grid):                    # var_state.py:88-91
        real_space_probs = self(torch.as_tensor(grid.coords)[None, ...], mode="eval")
        return torch.sum(grid.bin_area * torch.exp(real_space_probs))

    # ---- parameter utilities, var_state.py:94-108 ---------------------------------------------------------------
    def set_parameters(self, p_new):
        self._theta = torch.as_tensor(p_new, dtype=global_defs.tReal, device=self.device).reshape(-1).clone()
        assert self._theta.numel() == self.numParameters
        self._params = None

    @property
    def params(self):
        """var_state.py:94-101: the nested {"params": {...}} dict of views into the flat vector (read by main.py:178,190 and
        passed to net.apply by main.py:81-95).  Built on first access after set_parameters -- ~70 view objects, 0.2 ms of host
        time that a TDVP call itself never needs."""
        if self._params is None:
            self._params = self.net.tree_from_flat(self._theta)
        return self._params

    @params.setter
    def params(self, tree):
        self._theta = self.net.flat_from_tree(tree).clone()
        self._params = None

    def get_parameters(self):
        return self._theta.clone()

    def flatten_tree(self, tree):
        return self.net.flat_from_tree(tree)

    def init_net(self, network_args, key, depth, inds_up=None, inds_down=None, **kwargs):
        """var_state.py:110-124: per block a random d//2 subset (order kept) and its sorted complement.
        (numpy Generator instead of the JAX key; partitions can also be injected.)"""
        if inds_up is None:
            rng = np.random.default_rng(int(key))
            inds_up, inds_down = [], []
            for _ in range(depth):
                up = [int(v) for v in rng.permutation(self.dim)[: self.dim // 2]]
                inds_up.append(up)
                inds_down.append([i for i in range(self.dim) if i not in up])
        mynet = net.INNwProb(inds_up, inds_down, device=self.device, **network_args)
        theta = mynet.init(int(key))
        return mynet, theta
