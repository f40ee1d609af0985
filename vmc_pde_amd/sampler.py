"""Latent-space samplers (reference: vmc_fluids/sampler.py).

Exact sampling for "Gauss"/"Student_t" (sampler.py:25-34) and the independence-Metropolis MCMC for any other
name (sampler.py:88-111, dormant and broken in the reference at this commit -- `latent_space_prob` is undefined
there; the target is the `cos_dist` of sampler.py:20-22 as its `__main__` block and main.py:33 intend).
Random numbers come from the counter-based Philox generator in csrc/sampler.hip: JAX's threefry stream and the
reference's unseeded numpy chisquare cannot be reproduced; parity tests inject the draws instead.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _native as nat
from . import global_defs
from . import mpi_wrapper


def radial_update_prop(key, numChains, mcmc_info):
    """sampler.py:37-45 on the host (used only to initialise the chains, sampler.py:67-70)."""
    g = torch.Generator().manual_seed(int(key))
    r = torch.sqrt(torch.rand((numChains, 1), generator=g, dtype=torch.float64)) * mcmc_info["bound"]
    phi = 2 * math.pi * torch.rand((numChains, 1), generator=g, dtype=torch.float64)
    off = torch.as_tensor(np.asarray(mcmc_info["offset"], dtype=np.float64))
    return torch.cat((r * torch.cos(phi), r * torch.sin(phi)), dim=-1) + off


class Sampler:
    """sampler.py:48-65.  `key` is an integer seed; every call consumes one Philox stream id (the analogue of the
    reference's `jax.random.split` per call, sampler.py:73).  Sample n of a call is a pure function of
    (key, call number, global sample index n), so sharding the samples over ranks reproduces the same global set."""

    def __init__(self, key=0, numChains=1, dim=2, name="Gauss", updateProposer=radial_update_prop, mcmc_info=None):
        self.key = int(key)
        self.numChains = numChains
        self.dim = dim
        self.name = name
        self.updateProposer = updateProposer
        self.mcmc_info = mcmc_info
        self.exact_samples = self.name in ["Gauss", "Student_t"]   # sampler.py:61
        self.states = None
        self._calls = 0
        self.device = global_defs.my_device()
        self.numAccepted = None

    def _next_stream(self):
        self._calls += 1
        return self._calls

    def mc_init(self):
        """sampler.py:67-70."""
        self.states = self.updateProposer(self.key * 7919 + 13 + mpi_wrapper.get_rank(), self.numChains, self.mcmc_info)[None, ...].to(self.device)

    def __call__(self, numSamples, dist_params, multipleOf=1):
        """sampler.py:72-86.  Returns latent samples (1, n_local, dim), already shifted by mcmc_info["offset"]."""
        stream_id = self._next_stream()
        # (the offset stays on the host until a branch needs it on the device: an upload from pageable memory is a blocking copy,
        # i.e. a host synchronisation per call, and the native exact sampler reads the offset from the model descriptor)
        offset_host = np.asarray(self.mcmc_info["offset"], dtype=np.float64).reshape(-1)
        if not self.exact_samples:
            if self.states is None:
                self.mc_init()
            steps = mpi_wrapper.distribute_sampling(numSamples, localDevices=global_defs.device_count(),
                                                    numChainsPerDevice=int(np.lcm(self.numChains, multipleOf)))
            chains = self.states.shape[1]
            states = self.states[0].clone().contiguous()   # the reference never writes self.states back
            out = torch.empty((steps * chains, 2), dtype=global_defs.tReal, device=self.device)
            n_acc = torch.zeros(1, dtype=torch.int64, device=self.device)
            off2 = (C.c_double * 2)(float(offset_host[0]), float(offset_host[1]))
            with torch.cuda.device(self.device):
                nat.call("vmcpde_mcmc", self.key, stream_id + (mpi_wrapper.get_rank() << 20), chains, steps,
                         float(self.mcmc_info["bound"]), off2, nat.ptr(states), nat.ptr(out), nat.ptr(n_acc), nat.stream())
            self.numAccepted = n_acc
            return out[None, ...]
        # exact samplers
        n_local = mpi_wrapper.distribute_sampling(numSamples)
        first = mpi_wrapper.first_sample_id(numSamples)
        if "_model" in dist_params:     # fused native path (VarState.sample): z = mu + chol(L L^T) n (* t-scale) + offset
            model, theta = dist_params["_model"], dist_params["_theta"]
            z = torch.empty((n_local, self.dim), dtype=global_defs.tReal, device=self.device)
            with torch.cuda.device(self.device):
                nat.call("vmcpde_sample_latent", model.handle, nat.ptr(theta), self.key, stream_id, first, n_local, nat.ptr(z),
                         nat.stream())
            # the kernel adds the NETWORK's offset (one descriptor); sampler.py:86 adds mcmc_info["offset"] -- the same vector in
            # the reference's driver (main.py:69-73); if a caller makes them differ, the difference is added here
            d_off = offset_host - np.asarray(model.offset, dtype=np.float64).reshape(-1)
            if np.any(d_off != 0.0):
                z += torch.as_tensor(d_off, device=self.device)
            return z[None, ...]
        # generic (S, mu) given by the caller: small torch ops, same law (sampler.py:26,33-34)
        g = torch.Generator(device=self.device).manual_seed(self.key * 1000003 + stream_id * 4099 + mpi_wrapper.get_rank())
        S, mu = dist_params["S"].to(self.device), dist_params["mu"].to(self.device)
        y = torch.randn((n_local, self.dim), generator=g, dtype=global_defs.tReal, device=self.device) @ torch.linalg.cholesky(S).T
        if self.name == "Gauss":
            z = mu + y
        else:
            nu = float(torch.exp(dist_params["dist_params"][0]) + 1.0)
            u = torch.distributions.Chi2(torch.tensor(nu, dtype=global_defs.tReal)).sample((n_local,)).to(self.device)
            z = torch.sqrt(nu / u)[:, None] * y + mu
        return (z + torch.as_tensor(offset_host, device=self.device))[None, ...]
