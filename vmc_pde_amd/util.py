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


def store_infos(wdir, infos, name=None):
    """Run record of the driver loop (util.py:29-32): every key of `infos` becomes one array stacked over the steps.
    `name` ending in ".hdf5" writes the reference's own file -- one HDF5 dataset per key, what its plotting scripts open
    (visualization.py:285) -- and needs h5py; ".npz" writes the same arrays with numpy (`np.load(...)[key]` replaces
    `h5py.File(...)[key][:]`).  Default: "infos.hdf5" where h5py is importable, else "infos.npz" (h5py is not part of this image).
    Either file replaces the previous one atomically.  Returns the path written."""
    def as_numpy(v):
        return v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    arrays = {key: np.asarray([as_numpy(v) for v in series]) for key, series in infos.items()}
    h5py = None
    if name is None or name.endswith((".hdf5", ".h5")):
        try:
            import h5py
        except ImportError:
            if name is not None:
                raise ImportError(f"store_infos(name={name!r}) needs h5py; pass a name ending in .npz or install h5py") from None
    if name is None:
        name = "infos.hdf5" if h5py is not None else "infos.npz"
    target = os.path.join(wdir, name)
    if h5py is not None:
        tmp = target + ".tmp"
        with h5py.File(tmp, "w") as f:
            for key, value in arrays.items():
                f.create_dataset(key, data=value)
    else:
        tmp = target + ".tmp.npz"          # np.savez appends .npz to names without it
        np.savez(tmp, **arrays)
    os.replace(tmp, target)
    return target


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
