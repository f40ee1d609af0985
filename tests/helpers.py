"""Shared test helpers: one network configuration -> (oracle spec, theta) and the matching product VarState."""
import numpy as np
import torch

from oracle import vmc_oracle as O


def make_spec(dim, depth, hidden, latent="Gauss", coupling="no_add", seed=1, offset=None):
    ups, dns = O.random_partitions(dim, depth, seed)
    off = np.zeros(dim) if offset is None else np.asarray(offset, dtype=np.float64)
    global_change = coupling.endswith("+global")          # e.g. "no_add+global": net.py:72
    coupling = coupling.split("+")[0]
    flags = dict(jac_eq_1=coupling == "jac_eq_1", different_add=coupling == "different_add",
                 no_add=coupling in ("no_add",), global_change=global_change)
    return O.NetSpec(dim=dim, inds_up=ups, inds_down=dns, intmediate=tuple(hidden), offset=off.tolist(),
                     latentSpaceName=latent, **flags)


def random_theta(spec, seed, out_scale=0.05, latent_scale=0.3):
    """Parameters with every block active (the reference's 1e-5 output scale makes the flow ~identity)."""
    th = O.init_parameters(spec, seed, out_scale=out_scale)
    g = torch.Generator().manual_seed(seed + 77)
    p = O.unflatten(spec, th.clone())
    n_lat = p["L"].numel() + 2 * spec.dim + spec.n_dist
    lat = latent_scale * (torch.rand(n_lat, generator=g, dtype=torch.float64) - 0.5)
    # latent leaves are the first n_lat entries of the flat vector (L, L_diag, dist_params, mu)
    th[:n_lat] = lat
    # biases too
    for name, shape in O.param_layout(spec):
        pass
    off = 0
    for name, shape in O.param_layout(spec):
        n = int(np.prod(shape))
        if name.endswith("/bias") or name.endswith("/global_offset"):
            th[off:off + n] = 0.2 * (torch.rand(n, generator=g, dtype=torch.float64) - 0.5)
        elif name.endswith("/global_scale"):
            th[off:off + n] = 1.0 + 0.4 * (torch.rand(n, generator=g, dtype=torch.float64) - 0.5)
        off += n
    return th


def make_varstate(spec, theta, sampler=None, jacobian_dtype=None):
    import vmc_pde_amd as V
    coupling = dict(jac_eq_1=spec.jac_eq_1, different_add=spec.different_add, no_add=spec.no_add, global_change=spec.global_change)
    if sampler is None:
        sampler = V.Sampler(key=1, dim=spec.dim, name=spec.latentSpaceName, mcmc_info={"offset": np.asarray(spec.offset), "bound": 0.25})
    vs = V.VarState(sampler, spec.dim, 1, spec.depth,
                    network_args=dict(intmediate=spec.intmediate, offset=np.asarray(spec.offset), latentSpaceName=spec.latentSpaceName,
                                      dim=spec.dim, **coupling),
                    inds_up=spec.inds_up, inds_down=spec.inds_down,
                    **({} if jacobian_dtype is None else {"jacobian_dtype": jacobian_dtype}))
    vs.set_parameters(theta)
    return vs


class FixedSampler:
    """Duck-typed sampler that returns injected latent samples (parity tests)."""

    def __init__(self, z, key=1):
        self.z = z
        self.key = key

    def __call__(self, numSamples, dist_params, multipleOf=1):
        return self.z


def rel_err(a, b):
    a = a.detach().cpu().double() if torch.is_tensor(a) else torch.as_tensor(a, dtype=torch.float64)
    b = b.detach().cpu().double() if torch.is_tensor(b) else torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))
