"""Device registry and dtype aliases (reference: vmc_fluids/global_defs.py).

The reference keeps a list of pmap devices plus an MPI rank -> device map (global_defs.py:16-24).  Here there is
one process per GPU: the device is `cuda:LOCAL_RANK`, the "device axis" of every batched array has size 1.
"""
import os

import torch

tReal = torch.float64      # global_defs.py:8
tCpx = torch.complex128    # global_defs.py:6


def local_rank() -> int:
    return int(os.environ.get("LOCAL_RANK", "0"))


def my_device() -> torch.device:   # global_defs.py:16-20
    return torch.device("cuda", local_rank() % max(1, torch.cuda.device_count()))


def device_count() -> int:         # global_defs.py:47-48: one device per process
    return 1
