"""Data-parallel reductions over Monte-Carlo samples (reference: vmc_fluids/mpi_wrapper.py).

The reference does pmap psum + host-buffer MPI.Allreduce around every mean (10 all-reduces per TDVP step, two of
them P x P, each with a device->host->device round trip; mpi_wrapper.py:129-274).  Here ranks are torch.distributed
processes (one per GPU, backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests) and the TDVP engine packs all
partial sums of a step into ONE buffer that is all-reduced once (tdvp.py in this package).  The functions below keep
the reference names for the generic cases.
"""
import os as _os

import torch
import torch.distributed as dist

globNumSamples = 0          # mpi_wrapper.py:13, set by TDVP.__call__ (tdvp.py:99)


def _on():
    return dist.is_available() and dist.is_initialized()


def get_rank():
    return dist.get_rank() if _on() else 0


def get_comm_size():
    return dist.get_world_size() if _on() else 1


def __getattr__(name):      # `mpi.rank` / `mpi.commSize` as module attributes (mpi_wrapper.py:9-11)
    if name == "rank":
        return get_rank()
    if name == "commSize":
        return get_comm_size()
    raise AttributeError(name)


def _shard_sizes(total, parts):
    """Sizes of `parts` contiguous shards of `total` items: the first `total % parts` shards hold one item more."""
    base, extra = divmod(int(total), int(parts))
    return [base + (r < extra) for r in range(parts)]


def distribute_sampling(numSamples, localDevices=None, numChainsPerDevice=1):
    """mpi_wrapper.py:68-110.  Exact samplers (localDevices None): this rank's shard of `numSamples`.  MCMC: every rank runs
    `localDevices * numChainsPerDevice` chains, so the return value is the number of sweeps that covers its shard, and the global
    sample count is rounded up to whole sweeps on every rank.  Sets `globNumSamples` either way."""
    global globNumSamples
    shards = _shard_sizes(numSamples, get_comm_size())
    if localDevices is None:
        globNumSamples = numSamples
        return shards[get_rank()]
    chains = localDevices * numChainsPerDevice
    sweeps = [-(-n // chains) for n in shards]
    globNumSamples = chains * sum(sweeps)
    return sweeps[get_rank()]


def first_sample_id(numSamples=None):
    """mpi_wrapper.py:113-126: global index of this rank's first sample = the sizes of the shards before it."""
    total = globNumSamples if numSamples is None else numSamples
    return sum(_shard_sizes(total, get_comm_size())[:get_rank()])


_abi_comm = None       # ncclComm_t made through the C ABI (vmcpde_comm_init_rank), one per process


def abi_communicator():
    """RCCL communicator owned by libvmcpde: rank 0 draws the 128-byte id (vmcpde_comm_unique_id), the torch.distributed
    rendezvous that is up anyway carries it to the other ranks, and every rank binds its current GPU
    (vmcpde_comm_init_rank).  A reference-side binder would do the same with an mpi4py bcast."""
    global _abi_comm
    if _abi_comm is None:
        import ctypes as C
        from . import _native as nat
        ident = (C.c_char * 128)()
        if get_rank() == 0:
            nat.call("vmcpde_comm_unique_id", C.cast(ident, C.c_void_p))
        box = [bytes(ident.raw)]
        dist.broadcast_object_list(box, src=0)
        ident = (C.c_char * 128).from_buffer_copy(box[0])
        comm = C.c_void_p()
        nat.call("vmcpde_comm_init_rank", C.byref(comm), get_comm_size(), C.cast(ident, C.c_void_p), get_rank())
        _abi_comm = comm
    return _abi_comm


def destroy_abi_communicator():
    global _abi_comm
    if _abi_comm is not None:
        from . import _native as nat
        nat.call("vmcpde_comm_destroy", _abi_comm)
        _abi_comm = None


def allreduce_sum_(buf: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks of one packed device buffer (the single collective of a TDVP step).
    backend "nccl" = RCCL over xGMI, either through torch.distributed (default) or -- VMCPDE_COLLECTIVE=abi -- through the
    library's own entry point `vmcpde_allreduce` on a communicator made by `abi_communicator` (the path a non-torch host
    would bind).  backend "gloo": CPU rehearsal of the multi-rank logic, staged through the host."""
    if _on() and dist.get_world_size() > 1:
        if buf.is_cuda and dist.get_backend() == "gloo":
            tmp = buf.cpu()
            dist.all_reduce(tmp, op=dist.ReduceOp.SUM)
            buf.copy_(tmp)
        elif buf.is_cuda and buf.dtype == torch.float64 and buf.is_contiguous() and _os.environ.get("VMCPDE_COLLECTIVE", "torch") == "abi":
            from . import _native as nat
            with torch.cuda.device(buf.device):
                nat.call("vmcpde_allreduce", nat.ptr(buf), buf.numel(), abi_communicator(), nat.stream())
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf


def allreduce_max_(buf: torch.Tensor) -> torch.Tensor:
    """In-place maximum over ranks of a small device buffer (a figure every rank must decide on alike)."""
    if _on() and dist.get_world_size() > 1:
        if buf.is_cuda and dist.get_backend() == "gloo":
            tmp = buf.cpu()
            dist.all_reduce(tmp, op=dist.ReduceOp.MAX)
            buf.copy_(tmp)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.MAX)
    return buf


def global_sum(data):
    """mpi_wrapper.py:129-163: sum over the device and batch axes and over ranks."""
    res = data.sum(dim=(0, 1))
    return allreduce_sum_(res.contiguous())


def global_mean(data):
    """mpi_wrapper.py:166-193."""
    return global_sum(data) / globNumSamples


def global_variance(data):
    """mpi_wrapper.py:196-245."""
    mean = global_mean(data)
    return global_sum((data - mean).abs() ** 2) / globNumSamples


def global_covariance(data):
    """mpi_wrapper.py:248-274: <X^T X> of already centred data (generic torch path; the TDVP engine uses the MFMA
    syrk kernel on the parameter-major Jacobian instead)."""
    x = data[0]
    return allreduce_sum_((x.T @ x).contiguous()) / globNumSamples
