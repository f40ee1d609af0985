"""Test helpers that touch only the product package (no oracle import)."""
import numpy as np
import torch


def rel_err(a, b):
    a = torch.as_tensor(np.asarray(a.detach().cpu()) if isinstance(a, torch.Tensor) else np.asarray(a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b.detach().cpu()) if isinstance(b, torch.Tensor) else np.asarray(b), dtype=torch.float64)
    return float(torch.linalg.norm((a.reshape(-1) - b.reshape(-1))) / (torch.linalg.norm(b.reshape(-1)) + 1e-300))


class FixedSampler:
    """Duck-typed sampler that returns injected latent samples (1,N,d)."""
    key = 1

    def __init__(self, z):
        self.z = z

    def __call__(self, numSamples, dist_params, multipleOf=1):
        return self.z


def build_varstate(dim, depth, hidden, latent, coupling, offset, inds_up, inds_down, theta, sampler=None):
    import vmc_pde_amd as V
    global_change = coupling.endswith("+global")          # e.g. "no_add+global": net.py:72
    coupling = coupling.split("+")[0]
    flags = dict(jac_eq_1=coupling == "jac_eq_1", different_add=coupling == "different_add", no_add=coupling == "no_add",
                 global_change=global_change)
    if sampler is None:
        sampler = V.Sampler(key=1, dim=dim, name=latent, mcmc_info={"offset": np.asarray(offset), "bound": 0.25})
    vs = V.VarState(sampler, dim, 1, depth,
                    network_args=dict(intmediate=tuple(hidden), offset=np.asarray(offset), latentSpaceName=latent, dim=dim, **flags),
                    inds_up=[[int(v) for v in r] for r in np.asarray(inds_up)], inds_down=[[int(v) for v in r] for r in np.asarray(inds_down)])
    vs.set_parameters(theta)
    return vs
