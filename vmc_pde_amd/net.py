"""Coupling-flow density model (reference: vmc_fluids/net.py).

`INNwProb` here is a thin host object: it owns the libvmcpde model handle (the device-side "flow program" with the
flat-parameter layout) and exposes the reference's `apply(params, x, evaluate, inv)` protocol (net.py:209-217).
All arithmetic happens in the HIP kernels of csrc/flow.hip.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from . import _native as nat
from . import global_defs


def _trafo_names(different_add: bool):
    return ["s1", "s2", "t1", "t2"] if different_add else ["s1", "s2"]   # net.py:75-79


class INNwProb:
    """net.py:185-217.  Constructor arguments mirror the flax module fields; the SingleBlock flags of net.py:69-72
    (which the reference can only change by editing net.py, main.py:43-58) are ordinary keyword arguments here."""

    def __init__(self, inds_up, inds_down, intmediate=(3,), offset=None, latentSpaceName="Gauss", dim=2,
                 jac_eq_1=False, different_add=False, no_add=True, global_change=False, device=None):
        self.inds_up = [[int(v) for v in u] for u in inds_up]
        self.inds_down = [[int(v) for v in u] for u in inds_down]
        self.intmediate = tuple(int(v) for v in intmediate)
        self.dim = int(dim)
        self.depth = len(self.inds_up)
        self.latentSpaceName = latentSpaceName
        self.offset = np.zeros(self.dim) if offset is None else np.asarray(
            offset.detach().cpu().numpy() if torch.is_tensor(offset) else offset, dtype=np.float64).reshape(-1)
        assert self.offset.shape[0] == self.dim
        if jac_eq_1:
            self.coupling = "jac_eq_1"        # net.py:90-92
        elif different_add:
            self.coupling = "different_add"   # net.py:93-94
        elif no_add:
            self.coupling = "no_add"          # net.py:95-96
        else:
            self.coupling = "plain"           # net.py:97-98
        self.global_change = bool(global_change)   # net.py:72,80-82: per-block y -> global_scale * y + global_offset
        self.device = global_defs.my_device() if device is None else torch.device(device)

        n_up = len(self.inds_up[0])
        assert all(len(u) == n_up for u in self.inds_up)
        self._keep = dict(
            up=(C.c_int32 * (self.depth * n_up))(*[v for u in self.inds_up for v in u]),
            dn=(C.c_int32 * (self.depth * (self.dim - n_up)))(*[v for u in self.inds_down for v in u]),
            off=(C.c_double * self.dim)(*self.offset.tolist()))
        desc = nat.ModelDesc()
        if len(self.intmediate) > len(desc.hidden):
            raise ValueError(f"n_hidden {len(self.intmediate)} outside [0,{len(desc.hidden)}] (VMCPDE_MAX_HIDDEN_LAYERS)")
        desc.dim, desc.depth, desc.n_hidden = self.dim, self.depth, len(self.intmediate)
        for k, w in enumerate(self.intmediate):
            desc.hidden[k] = w
        desc.coupling, desc.latent, desc.n_up = nat.COUPLING[self.coupling], nat.LATENT[latentSpaceName], n_up
        desc.global_change = 1 if self.global_change else 0
        desc.inds_up = C.cast(self._keep["up"], C.POINTER(C.c_int32))
        desc.inds_down = C.cast(self._keep["dn"], C.POINTER(C.c_int32))
        desc.offset = C.cast(self._keep["off"], C.POINTER(C.c_double))
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            nat.call("vmcpde_create", C.byref(desc), C.byref(h))
        self.handle = h
        self.numParameters = nat.lib().vmcpde_num_parameters(h)
        self.layout = self._build_layout()
        assert sum(int(np.prod(s)) for _, _, s in self.layout) == self.numParameters

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                nat.lib().vmcpde_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ---- flat parameter layout: JAX pytree order = sorted keys (var_state.py:25-27, 94-108) ------------------
    def _build_layout(self) -> List[Tuple[Tuple[str, ...], int, Tuple[int, ...]]]:
        """[(path, offset, shape)] in flat order; offsets come from the native layout table so that host and
        kernels cannot disagree."""
        d, L = self.dim, nat.lib()
        out = [(("L",), L.vmcpde_param_offset(self.handle, -1, 0, 0, 0), ((d * d - d) // 2,)),
               (("L_diag",), L.vmcpde_param_offset(self.handle, -1, 1, 0, 0), (d,))]
        n_dist = 1 if self.latentSpaceName == "Student_t" else 0
        off_mu = L.vmcpde_param_offset(self.handle, -1, 3, 0, 0)
        out.append((("dist_params",), off_mu - n_dist, (n_dist,)))
        out.append((("mu",), off_mu, (d,)))
        names = _trafo_names(self.coupling == "different_add")
        tr_index = {"s1": 0, "s2": 1, "t1": 2, "t2": 3}
        n_up = len(self.inds_up[0])
        for bn in sorted(f"blocks_{b}" for b in range(self.depth)):
            b = int(bn.split("_")[1])
            if self.global_change:
                out.append((("myINN", bn, "global_offset"), L.vmcpde_param_offset(self.handle, b, 4, 0, 0), (d,)))
                out.append((("myINN", bn, "global_scale"), L.vmcpde_param_offset(self.handle, b, 5, 0, 0), (1,)))
            for tn in names:
                n_in, n_out = (n_up, d - n_up) if tn in ("s1", "t1") else (d - n_up, n_up)
                widths = [n_in] + list(self.intmediate) + [n_out]
                for k in range(len(widths) - 1):
                    ob = L.vmcpde_param_offset(self.handle, b, tr_index[tn], k, 0)
                    ok = L.vmcpde_param_offset(self.handle, b, tr_index[tn], k, 1)
                    out.append((("myINN", bn, tn, f"Dense_{k}", "bias"), ob, (widths[k + 1],)))
                    out.append((("myINN", bn, tn, f"Dense_{k}", "kernel"), ok, (widths[k], widths[k + 1])))
        out.sort(key=lambda e: e[1])
        pos = 0
        for _, off, shape in out:   # the native table must tile [0, P) without gaps in this order
            assert off == pos, "host/native parameter layout mismatch"
            pos += int(np.prod(shape))
        return out

    def tree_from_flat(self, theta: torch.Tensor) -> Dict:
        """var_state.py:94-101: nested {"params": {...}} dict of VIEWS into the flat vector."""
        tree: Dict = {}
        for path, off, shape in self.layout:
            n = int(np.prod(shape))
            node = tree
            for key in path[:-1]:
                node = node.setdefault(key, {})
            node[path[-1]] = theta[off:off + n].view(shape)
        return {"params": tree}

    def flat_from_tree(self, params: Dict) -> torch.Tensor:
        """var_state.py:103-108."""
        leaves = []
        for path, _, _ in self.layout:
            node = params["params"]
            for key in path:
                node = node[key]
            leaves.append(torch.as_tensor(node, dtype=global_defs.tReal, device=self.device).reshape(-1))
        return torch.cat(leaves)

    def init(self, seed: int, out_scale: float = 1e-5, hidden_scale: float = 1.0) -> torch.Tensor:
        """net.py:39-41,48-49,55-56,60-61,201-204: hidden kernels U(-1,1), output kernels U(-1e-5,1e-5), zeros elsewhere.
        (JAX's threefry stream is not reproducible; a seeded torch generator stands in)"""
        g = torch.Generator().manual_seed(int(seed))
        theta = torch.zeros(self.numParameters, dtype=global_defs.tReal)
        n_dense = len(self.intmediate) + 1
        for path, off, shape in self.layout:
            if path[-1] == "kernel":
                k = int(path[-2].split("_")[1])
                scale = out_scale if k == n_dense - 1 else hidden_scale
                n = int(np.prod(shape))
                theta[off:off + n] = 2 * scale * (torch.rand(n, generator=g, dtype=global_defs.tReal) - 0.5)
            elif path[-1] == "global_scale":
                theta[off] = 1.0                       # net.py:81 initializers.ones
        return theta.to(self.device)

    def _latent_logpdf(self, theta, pts):
        """net.py:10-36 in torch ops on the device (Gauss incl. log det S, Student_t without it -- as the reference writes them) for
        points (n, d): latent_space_dist(pts - offset, {S = U U^T (util.py:21-26), mu, dist_params})."""
        tree, d = self.tree_from_flat(theta)["params"], self.dim
        U = torch.zeros((d, d), dtype=global_defs.tReal, device=self.device)
        iu = torch.triu_indices(d, d, offset=1, device=self.device)
        U[iu[0], iu[1]] = tree["L"]
        U = U + torch.diag(torch.exp(tree["L_diag"]))
        S = U @ U.T
        r = pts - torch.as_tensor(self.offset, dtype=global_defs.tReal, device=self.device) - tree["mu"]
        q = ((r @ torch.linalg.inv(S)) * r).sum(-1)
        if self.latentSpaceName == "Gauss":
            return -0.5 * (d * float(np.log(2 * np.pi)) + torch.logdet(S) + q)
        nu = torch.exp(tree["dist_params"][0]) + 1.0
        return (torch.lgamma((nu + d) / 2) - torch.lgamma(nu / 2) - d / 2 * torch.log(nu * float(np.pi)) - (nu + d) / 2 * torch.log1p(q / nu))

    # ---- net.py:209-217 ---------------------------------------------------------------------------------------
    def apply(self, params, x, evaluate=True, inv=False):
        """Single sample (d,) or batch (N,d).  evaluate=True: log p(x) (inv=True: the reference's formula with the inverse walk, net.py:209-213);
        evaluate=False: (y, latent(x - offset) - log_jac) in either direction."""
        theta = params if torch.is_tensor(params) else self.flat_from_tree(params)
        theta = theta.to(self.device, global_defs.tReal).contiguous()
        xb = torch.as_tensor(x, dtype=global_defs.tReal, device=self.device)
        single = xb.dim() == 1
        xb = xb.reshape(-1, self.dim).contiguous()
        n = xb.shape[0]
        with torch.cuda.device(self.device):
            if evaluate:
                if inv:
                    # net.py:209-213 passes `inv` through: z, log_jac = INN(x, inv=True), result = latent(z - offset) + log_jac.  None of
                    # the reference's callers uses it; the inverse walk comes from the sampling kernel (which returns
                    # latent(x - offset) - log_jac), the two latent densities from the closed forms of net.py:10-36
                    y = torch.empty_like(xb)
                    lp = torch.empty(n, dtype=global_defs.tReal, device=self.device)
                    nat.call("vmcpde_flow_sample", self.handle, nat.ptr(theta), nat.ptr(xb), n, 1, nat.ptr(y), nat.ptr(lp), nat.stream())
                    out = self._latent_logpdf(theta, y) + (self._latent_logpdf(theta, xb) - lp)
                    return out[0] if single else out
                logp = torch.empty(n, dtype=global_defs.tReal, device=self.device)
                nat.call("vmcpde_flow_eval", self.handle, nat.ptr(theta), nat.ptr(xb), n, None, 0, nat.ptr(logp), None, None,
                         None, 0, 0, None, 0, nat.stream())
                return logp[0] if single else logp
            y = torch.empty_like(xb)
            logp = torch.empty(n, dtype=global_defs.tReal, device=self.device)
            nat.call("vmcpde_flow_sample", self.handle, nat.ptr(theta), nat.ptr(xb), n, 1 if inv else 0, nat.ptr(y), nat.ptr(logp),
                     nat.stream())
        return (y[0], logp[0]) if single else (y, logp)
