"""Helpers (reference: vmc_fluids/util.py)."""
import os
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
    """Run record of the driver loop (util.py:29-32): every key of `infos` becomes one array stacked over the steps.
    Written as .npz -- h5py is not part of the image; `np.load(...)[key]` replaces `h5py.File(...)[key][:]`."""
    def as_numpy(v):
        return v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    target = os.path.join(wdir, name)
    tmp = target + ".tmp.npz"          # np.savez appends .npz to names without it
    np.savez(tmp, **{key: np.asarray([as_numpy(v) for v in series]) for key, series in infos.items()})
    os.replace(tmp, target)


class Timings:
    """Named wall-clock sections (util.py:35-52): `timing_dict[name]` is the list of durations of that section, one per
    occurrence.  The caller synchronises the device at the section boundaries (TDVP does, like block_until_ready)."""

    def __init__(self):
        self.timing_dict = {}
        self._open = {}

    def start_timing(self, key):
        self._open[key] = time.perf_counter()

    def stop_timing(self, key):
        self.timing_dict.setdefault(key, []).append(time.perf_counter() - self._open.pop(key))

    def print_timings(self):
        latest = {key: series[-1] for key, series in self.timing_dict.items() if series}
        for key, seconds in latest.items():
            print(f"\t > {key}: {seconds}")
        print(f"\t > TOTAL: {sum(latest.values())}")
