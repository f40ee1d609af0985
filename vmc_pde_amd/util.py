"""Helpers (reference: vmc_fluids/util.py)."""
import time

import numpy as np
import torch

from . import global_defs


def build_cov_matrix(L_para, L_diag, dim):
    """util.py:21-26: strict upper triangle (row-major) + diag(exp(L_diag)); returns L L^T.  Small d x d torch op;
    the kernels rebuild the same factor on device (flow.hip latent_prepare_kernel)."""
    L = torch.zeros((dim, dim), dtype=global_defs.tReal, device=L_diag.device)
    iu = torch.triu_indices(dim, dim, offset=1, device=L_diag.device)
    if L_para.numel():
        L[iu[0], iu[1]] = L_para
    L = L + torch.diag(torch.exp(L_diag))
    return L @ L.T


def store_infos(wdir, infos, name="infos.npz"):
    """util.py:29-32 (one dataset per key; .npz because h5py is not part of the image)."""
    out = {}
    for key, value in infos.items():
        out[key] = np.asarray([v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v) for v in value])
    np.savez(wdir + name, **out)


class Timings:
    """util.py:35-52: wall-clock sections; stop_timing synchronises the device like block_until_ready."""

    def __init__(self):
        self.timing_dict = {}

    def start_timing(self, key):
        self.timing_dict.setdefault(key, []).append(-time.perf_counter())

    def stop_timing(self, key):
        self.timing_dict[key][-1] += time.perf_counter()

    def print_timings(self):
        total = 0
        for key, value in self.timing_dict.items():
            print(f"\t > {key}: {value[-1]}")
            total += value[-1]
        print(f"\t > TOTAL: {total}")
