"""Data-parallel reductions over Monte-Carlo samples (reference: vmc_fluids/mpi_wrapper.py).

The reference does pmap psum + host-buffer MPI.Allreduce around every mean (10 all-reduces per TDVP step, two of
them P x P, each with a device->host->device round trip; mpi_wrapper.py:129-274).  Here ranks are torch.distributed
processes (one per GPU, backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests) and the TDVP engine packs all
partial sums of a step into ONE buffer that is all-reduced once (tdvp.py in this package).  The functions below keep
the reference names for the generic cases.
"""
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


def distribute_sampling(numSamples, localDevices=None, numChainsPerDevice=1):
    """mpi_wrapper.py:68-110: samples per rank (or per chain when localDevices is given); sets globNumSamples."""
    global globNumSamples
    commSize, rank = get_comm_size(), get_rank()
    samplesPerProcess = numSamples // commSize
    if rank < numSamples % commSize:
        samplesPerProcess += 1
    if localDevices is None:
        globNumSamples = numSamples
        return samplesPerProcess
    numChainsPerProcess = localDevices * numChainsPerDevice

    def spc(spp):
        return (spp + numChainsPerProcess - 1) // numChainsPerProcess

    a = numSamples % commSize
    globNumSamples = (a * spc(1 + numSamples // commSize) + (commSize - a) * spc(numSamples // commSize)) * numChainsPerProcess
    return spc(samplesPerProcess)


def first_sample_id(numSamples=None):
    """mpi_wrapper.py:113-126: global index of this rank's first sample."""
    n = globNumSamples if numSamples is None else numSamples
    commSize, rank = get_comm_size(), get_rank()
    mySamples = n // commSize
    firstSampleId = rank * mySamples
    firstSampleId += rank if rank < n % commSize else n % commSize
    return firstSampleId


def allreduce_sum_(buf: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks of one packed device buffer (the single collective of a TDVP step)."""
    if _on() and dist.get_world_size() > 1:
        if buf.is_cuda and dist.get_backend() == "gloo":     # CPU-backend rehearsal of the multi-rank path: stage via host
            tmp = buf.cpu()
            dist.all_reduce(tmp, op=dist.ReduceOp.SUM)
            buf.copy_(tmp)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)       # backend "nccl" = RCCL over xGMI
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
