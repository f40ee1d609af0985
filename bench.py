#!/usr/bin/env python3
"""Headline benchmark: TDVP steps/s at N_samp = 65 536, N_param ~ 4k (BASELINE.json, config C2).

    python bench.py [--gpus N --steps K --warmup W] [--workload C2|C3|C4]

N > 1: one rank per GPU over RCCL.  Either under a launcher (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`: RANK / LOCAL_RANK / WORLD_SIZE are read from the environment) or
plainly as `python bench.py --gpus N`: this process then starts the N ranks itself as fresh child processes
(vmc_pde_amd/launch.py; the parent never touches the GPU), relays rank 0's JSON line and exits non-zero if any rank does.
The documented 8-GPU line is `python bench.py --gpus 8 --workload C4` (BASELINE config 4: 262 144 samples = 8 x 32 768, weak).

One "step" = one evaluation of TDVP.__call__ through `update` (tdvp.py:96-128: sample -> Eloc + Jacobian -> S, F ->
solve; observables excluded) on synthetic inputs: 2-D Fokker-Planck (Kramers equation with anharmonic
drift, evolutionEq.py `advection_hamiltonian_wDiss`, lam = 0.1), coupling flow d=2, 4 blocks, hidden (21,21) -> P = 4213
random-init parameters, fp64, latent draws from the on-device Philox sampler.  Weak scaling: every rank processes
65 536 samples; `value` = (samples/s over all ranks) / 65 536, i.e. TDVP steps/s in units of 65 536-sample steps
(equal to plain steps/s at N = 1).

The JSON line also carries
  roofline     -- the dominant kernel (fp64 MFMA Gram): algorithmic flops N*P*(P+1) per launch / average launch time,
                  measured with HIP events on the launch stream inside the timed steps; peak = 78.6 TFLOP/s (AMD's
                  MI355X fp64 matrix figure; the local guide lists no f64 MFMA rate) next to this run's measured
                  back-to-back v_mfma_f64_16x16x4_f64 rate.
  cpu_baseline -- the CPU oracle (oracle/vmc_oracle.py, the reference algorithm as written) timed on this box's host
                  cores on a bounded sample and scaled to the full step (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # multi-process GPU work on this pool needs dmabuf IPC (RCCL)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

N_SAMP = 65536
DIM, DEPTH, HIDDEN = 2, 4, (21, 21)
EQ = "advection_hamiltonian_wDiss"
LAM = 0.1
# BASELINE.json configs; the default (and the only one `value` is quoted on) is C2.  C3/C4 are functional / scaling records.
WORKLOADS = {
    "C2": dict(dim=2, hidden=(21, 21), n_samp=65536, offset=[1.0, 1.0],
               desc="C2: 2-D Fokker-Planck (Kramers, anharmonic drift lam=0.1), coupling flow d=2 x 4 blocks x hidden (21,21)"),
    "C3": dict(dim=2, hidden=(43, 43), n_samp=65536, offset=[1.0, 1.0],
               desc="C3: wide MLP, coupling flow d=2 x 4 blocks x hidden (43,43), P=16181"),
    # (harmonic: the reference's 'harmonicOsc_diff' mode has lam = 0; and a narrower output initialisation -- with U(-0.1, 0.1) the
    # random 6-D flow scales coordinates by up to e^10, Eloc reaches 1e13 and a Heun stage away from it is NaN)
    "C4": dict(dim=6, hidden=(19, 19), n_samp=32768, offset=[1.0, 0.0, 0.0, 1.0, 0.0, 0.0], lam=0.0, out_scale=0.02,
               desc="C4: 6-D Fokker-Planck (three damped oscillators, harmonic), coupling flow d=6 x 4 blocks x hidden (19,19), 32768 samples per GPU"),
}
DIAG_SHIFT = 1e-4                # a shift the headline net integrates stably with (tests/test_gpu_published_trajectory.py); timing does not depend on it
SVD_TOL = 1e-6                   # absolute floor svdTol * max diag(S): what keeps the over-parametrised headline net integrable
F32BLOCKS_DIAG_SHIFT = 1e-4     # --jacobian f32blocks only (vmc_pde_amd.tdvp.F32_BLOCK_MIN_SHIFT)
PEAK_F64_MFMA_TFLOPS = 78.6     # AMD MI355X spec, fp64 matrix (dense); the guide has no measured f64 MFMA figure
PEAK_F32_MFMA_TFLOPS = 157.3    # fp32 matrix (MI355X_MICROARCH.md: 157.3 spec / 155 measured) -- only for --jacobian f32blocks
PEAK_I8_MFMA_POPS = 5.0         # int8 matrix, dense (MI355X_MICROARCH.md: 2x the bf16 rate per clock) -- only for --jacobian i8sliced
DTYPE_LABEL = {"f64": "f64", "i8f32": "f32 Jacobian storage, four 8-bit int8 digits, int32 sums, f64 combine (BASELINE config 5 on the int8 pipe, experiment)",
               "i8sliced": "f64 Jacobian, error-free int8 slicing (six balanced 8-bit digits), int32 sums, f64 combine (experiment)",
               "f32": "f64 accumulate, f32 Jacobian storage (BASELINE config 5)",
               "f32blocks": "f32 Jacobian, f32 MFMA sums over 128-sample blocks, f64 across blocks (BASELINE config 5, fast variant)"}
OPERAND_LABEL = {"f64": "uncentred+DMA", "i8f32": "uncentred, 4 int8 digit planes + DMA", "i8sliced": "uncentred, int8 digit planes + DMA", "f32": "uncentred+DMA over fp32 sample pairs, widened at the fragment read", "f32blocks": "uncentred+DMA, fp32 MFMA blocks"}
KERNEL_LABEL = {"f64": "gram_dma_kernel (fp64 MFMA syrk, + split-K reduce)",
                "i8f32": "row_expo + slice + gram_i8_kernel<4,8,8> (v_mfma_i32_32x32x32_i8, 10 digit pairs) + split reduce + diagonal fix",
                "i8sliced": "row_expo + slice + gram_i8_kernel<6,8,8> (v_mfma_i32_32x32x32_i8, 21 digit pairs) + split reduce + diagonal fix",
                "f32": "gram_w32_kernel (fp32 operands DMA-staged as sample pairs, widened at the fragment read, fp64 MFMA syrk, + split-K reduce)",
                "f32blocks": "gram_f32_kernel (fp32 MFMA block sums, fp64 across blocks, + split-K reduce)"}
OUT_SCALE = 0.1                 # output kernels U(-0.1, 0.1) (the reference's 1e-5 makes J nearly degenerate)


def build_problem(device, workload="C2", seed=1, jacobian_dtype=torch.float64, experimental=False):
    import vmc_pde_amd as V
    w = WORKLOADS[workload]
    dim, off = w["dim"], np.asarray(w["offset"])
    sampler = V.Sampler(key=1234, dim=dim, name="Gauss", mcmc_info={"offset": off, "bound": 0.25})
    vs = V.VarState(sampler, dim, seed, DEPTH, network_args=dict(intmediate=w["hidden"], offset=off, latentSpaceName="Gauss", dim=dim),
                    jacobian_dtype=jacobian_dtype)
    theta = vs.net.init(seed, out_scale=w.get("out_scale", OUT_SCALE))
    vs.set_parameters(theta)
    eq = V.EvolutionEquation(dim=dim, name=EQ)
    eq.eqParams[EQ]["params"]["lam"] = w.get("lam", LAM)
    # (the product class; the opt-in experiments -- --jacobian i8sliced / i8f32, --solver matrixfree / eigh-matrixfree -- are a subclass
    # of it in vmc_pde_amd/experimental and are reported beside the headline, never as it)
    cls = V.TDVP
    if experimental:
        from vmc_pde_amd.experimental import TDVP as cls
    tdvp = cls(diagonalShift=DIAG_SHIFT, svdTol=SVD_TOL, solver="cholesky", fuseMoments=os.environ.get("VMCPDE_FUSE_MOMENTS", "1") != "0")
    return vs, eq, tdvp, theta


def beside_headline(device, args, kwargs, theta, steps=10, warmup=5):
    """TDVP steps/s of the opt-in experiments (never the headline): same workload, same regularised system, fp64 results (every call
    draws its own samples, as in the timed loop; parity of each variant with the assembled fp64 path on the SAME samples is
    tests/test_gpu_lowrank_eigh.py, tests/test_gpu_tdvp.py and tests/test_gpu_gram.py)."""
    import vmc_pde_amd as V
    from vmc_pde_amd.experimental import TDVP as XTDVP
    out = {"note": "measured after the timed region of this run; `value` is the assembled fp64 regularised-Cholesky path; the first two are opt-in EXPERIMENTS"}
    variants = {"matrixfree (S never assembled: TDVP(matrixFree=True), fp64)": dict(matrixFree=True),
                "int8Gram (error-free sliced S-build on the int8 matrix pipe: TDVP(int8Gram=True))": dict(int8Gram=True)}
    variants["TDVP() as the reference constructs it (tdvp.py:21-26: spectral pseudo-inverse, diagonalShift 0, svdTol 1e-11; not an experiment)"] = None
    for name, flags in variants.items():
        try:
            tv = XTDVP(diagonalShift=DIAG_SHIFT, svdTol=SVD_TOL, solver="cholesky", **flags) if flags is not None else V.TDVP()
            kw = dict(kwargs)
            for _ in range(warmup):
                u, _ = tv(theta, 0.0, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                u, _ = tv(theta, 0.0, **kw)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            tv.check("beside-headline variant")
            out[name] = {"steps_per_s": 1.0 / dt, "ms_per_step": dt * 1e3, "steps": steps, "solver_residual": float(tv.solverResidual)}
        except Exception as exc:        # an experiment that does not apply to this workload (a full-rank S) is reported, not fatal
            out[name] = {"error": f"{type(exc).__name__}: {str(exc)[:160]}"}
    return out


def collective_record(device, world, rank, backend, payload_bytes, reps=10):
    """What shows that `world` distinct devices took part in the step's collective (outside the timed region): the communicator's own
    size, every rank's PCI bus id (C ABI: hipDeviceGetPCIBusId of the device the rank computes on), a one-hot vector summed through
    the very path the step's all-reduce takes (every rank adds 1 at its own index: all ones = every rank contributed exactly once),
    and the packed buffer's all-reduce timed alone, back to back after a barrier (its exposed cost when nothing overlaps it)."""
    import ctypes as C
    from vmc_pde_amd import _native as nat
    from vmc_pde_amd import mpi_wrapper as mpi
    collective = "abi" if (backend == "nccl" and os.environ.get("VMCPDE_COLLECTIVE", "torch") == "abi") else "torch"
    buf = C.create_string_buffer(64)
    with torch.cuda.device(device):
        nat.call("vmcpde_device_bus_id", buf, 64)
    mine = {"rank": rank, "device_index": device.index, "bus_id": buf.value.decode(), "pid": os.getpid()}
    if collective == "abi":
        info = (C.c_int32 * 4)()
        nat.call("vmcpde_comm_info", mpi.abi_communicator(), info)
        mine["abi_comm"] = {"count": int(info[0]), "user_rank": int(info[1]), "hip_device": int(info[2]), "rccl_version": int(info[3])}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    onehot = torch.zeros(world, dtype=torch.float64, device=device)
    onehot[rank] = 1.0
    mpi.allreduce_sum_(onehot)
    pack = torch.zeros(payload_bytes // 8, dtype=torch.float64, device=device)
    for _ in range(2):
        mpi.allreduce_sum_(pack)
    dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        mpi.allreduce_sum_(pack)
    e1.record()
    torch.cuda.synchronize()
    alone_ms = e0.elapsed_time(e1) / reps
    t = torch.tensor([alone_ms], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    alone_ms = float(t.item())
    bus = [e["bus_id"] for e in everyone]
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "collective": collective,
            "ranks": everyone, "device_bus_ids": bus, "devices_distinct": len(set(bus)) == world,
            "onehot_sum": [float(v) for v in onehot.cpu()], "every_rank_contributed_once": bool((onehot == 1.0).all()),
            "allreduce_alone_ms": alone_ms, "allreduce_payload_bytes": payload_bytes,
            # ring all-reduce: each rank sends and receives 2 (N - 1) / N of the payload
            "allreduce_bus_GBps": 2.0 * (world - 1) / world * payload_bytes / (alone_ms * 1e-3) / 1e9}


def measure_mfma_peak(device):
    from vmc_pde_amd import _native as nat
    blocks, iters = 512, 4000      # iters is a multiple of 16 (the kernel's unroll)
    out = torch.empty(blocks * 256, dtype=torch.float64, device=device)
    nat.call("vmcpde_bench_mfma_f64", 200, blocks, nat.ptr(out), nat.stream())
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        nat.call("vmcpde_bench_mfma_f64", iters, blocks, nat.ptr(out), nat.stream())
        e1.record()
        torch.cuda.synchronize()
        flops = blocks * 4 * iters * 8 * 2048.0
        best = max(best, flops / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    return best


def _timed(fn, reps=2):
    """One untimed warm-up call, then `reps` timed calls: (median, min, all) in seconds."""
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), float(np.min(ts)), ts


def cpu_baseline_full(theta_cpu, inds_up, inds_down, workload="C2", slab=4096):
    """`--cpu-baseline full`: the reference algorithm as written (oracle), ONCE at the workload's real sample count -- no extrapolation.
    The per-sample autodiff (vmap(grad) + the Hessian pass) runs in slabs of `slab` samples whose Jacobians are concatenated into the
    N x P array the reference materialises (var_state.py:29-34; 2.2 GB at C2); everything behind it is tdvp.py:36-94 on the full
    arrays: centred copies, F0, the two un-symmetrised Gram GEMMs (S0, SExp), host eigh, EO @ V, pseudo-inverse.  One timed pass
    (~1 minute at C2 on 16 cores), so it is not part of the default run; the record goes to profiles/."""
    from oracle import vmc_oracle as O
    w = WORKLOADS[workload]
    dim, hidden, n_full = w["dim"], w["hidden"], w["n_samp"]
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    torch.set_num_threads(cores)
    try:
        from threadpoolctl import threadpool_limits
        blas_cap = threadpool_limits(limits=cores)
    except ImportError:
        blas_cap = None
    spec = O.NetSpec(dim=dim, inds_up=inds_up, inds_down=inds_down, intmediate=hidden, offset=list(w["offset"]))
    P = O.num_parameters(spec)
    if P > 8000:
        raise SystemExit("--cpu-baseline full: the host eigh of a 16 181 x 16 181 matrix does not fit a benchmark run (SURVEY 8d); use the default sample mode")
    vo = O.VarStateOracle(spec, theta_cpu)
    eo = O.EvolutionEquationOracle(dim, EQ)
    eo.eqParams[EQ]["params"]["lam"] = w.get("lam", LAM)
    g = torch.Generator().manual_seed(0)
    parts = {}
    t_all = time.perf_counter()
    t0 = time.perf_counter()
    Es, Js, Ls = [], [], []
    for first in range(0, n_full, slab):
        n = min(slab, n_full - first)
        z = O.latent_from_normals(spec, theta_cpu, torch.randn(n, dim, generator=g, dtype=torch.float64))
        x, _ = vo.flow_sample(z)
        E, J, lp = eo(vo, x, 0.0)
        Es.append(E), Js.append(J), Ls.append(lp)
    Eloc, J, logp = torch.cat(Es, 1), torch.cat(Js, 1), torch.cat(Ls, 1)
    del Es, Js, Ls
    parts["sample + autodiff J + Hessian (slabs of %d)" % slab] = time.perf_counter() - t0
    t0 = time.perf_counter()
    to = O.TDVPOracle()
    S, F, EOdata = to.get_tdvp_equation(Eloc, J, logp, n_full)          # tdvp.py:36-52
    parts["means, centred copies, F0, S0 and SExp GEMMs"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    ev, Vn = np.linalg.eigh(S.numpy())                                  # tdvp.py:61-64
    parts["host eigh"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    to.ev, to.V = torch.from_numpy(ev), torch.from_numpy(Vn)
    to.VtF = to.V.T @ F
    EO = EOdata[0] @ to.V                                               # tdvp.py:68
    rho = ((EO - EO.sum(0) / n_full) ** 2).sum(0) / n_full              # tdvp.py:70
    parts["EO @ V and its variance"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    rel = torch.abs(to.ev / to.ev[-1])
    upd = to.V @ (torch.where(rel > 1e-14, 1.0 / to.ev, torch.zeros_like(to.ev)) / (1.0 + (1e-11 / rel) ** 6) * to.VtF)   # tdvp.py:82-91
    resid = float(torch.linalg.norm(S @ upd - F) / torch.linalg.norm(F))
    parts["pseudo-inverse, residual"] = time.perf_counter() - t0
    total = time.perf_counter() - t_all
    if blas_cap is not None:
        blas_cap.restore_original_limits()
    return {"value": 1.0 / total, "unit": "TDVP steps/s", "cores": cores, "kind": "port", "workload": workload,
            "host_cpu_count": os.cpu_count() or 1, "affinity_count": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
            "sample": f"reference algorithm as written on torch/numpy fp64 CPU, ONE full step at N={n_full}, P={P}: nothing extrapolated",
            "seconds_per_step": total, "parts_seconds": parts, "solver_residual": resid, "rank_1e-9": int((rel > 1e-9).sum()),
            "note": "an eager torch.func port of the reference's algorithm, not XLA-jitted JAX-CPU (jax is not installable here): "
                    "a reported baseline, not a comparison with the reference's own runtime"}


def cpu_baseline(theta_cpu, inds_up, inds_down, workload="C2"):
    """The reference algorithm as written (oracle), timed piecewise on bounded samples -- one warm-up and two timed repetitions
    per part -- and extrapolated to the full step: the autodiff part (vmap(grad) + the Hessian pass) is timed at
    THREE sample counts and extrapolated along the fitted line t = a + b N (its cost per sample still falls between N = 2048 and
    4096, so a plain x16 of one point overstates it), the GEMM parts are scaled linearly in N, eigh is timed at the full P for C2.
    C3 (P = 16 181): autodiff and GEMMs as above; the host eigh of a 16 181 x 16 181 matrix does not fit a benchmark run (SURVEY 8d:
    "hours"), so it is timed at P/4 and scaled by 4^3 -- labelled as such."""
    from oracle import vmc_oracle as O
    w = WORKLOADS[workload]
    dim, hidden, n_full = w["dim"], w["hidden"], w["n_samp"]
    # threads actually used ("all host cores, count printed"): the host's count and this process's affinity are recorded, and the
    # pool width is the FASTEST of a sweep over 8 / 16 / 32 / 64 threads (those the affinity allows) on the largest autodiff point -- the
    # dominant part of the step.  An OpenMP pool as wide as the host's 256 hardware threads makes the oracle's many small ops slower,
    # not faster; the sweep in the record shows where that sets in on this host.
    host_cpus = os.cpu_count() or 1
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else host_cpus
    spec = O.NetSpec(dim=dim, inds_up=inds_up, inds_down=inds_down, intmediate=hidden, offset=list(w["offset"]))
    P = O.num_parameters(spec)
    vo = O.VarStateOracle(spec, theta_cpu)
    eo = O.EvolutionEquationOracle(dim, EQ)
    eo.eqParams[EQ]["params"]["lam"] = w.get("lam", LAM)
    g = torch.Generator().manual_seed(0)
    big = P > 8000
    n_gemm = 2048 if big else 4096
    ad_counts = (512, 1024, 2048) if big else (1024, 2048, 4096)
    candidates = sorted({t for t in (8, 16, 32, 64) if t <= affinity} | ({affinity} if affinity < 8 else set()))
    sweep = {}
    if len(candidates) > 1:
        zs = O.latent_from_normals(spec, theta_cpu, torch.randn(ad_counts[-1], dim, generator=g, dtype=torch.float64))

        def autodiff_sweep():
            x, _ = vo.flow_sample(zs)
            return eo(vo, x, 0.0)
        for t in candidates:
            torch.set_num_threads(t)
            sweep[t] = _timed(autodiff_sweep, 1)[0]
        cores = min(sweep, key=sweep.get)
    else:
        cores = candidates[0]
    torch.set_num_threads(cores)
    try:                                           # numpy's OpenBLAS pool is separate from torch's: same thread cap
        from threadpoolctl import threadpool_limits
        blas_cap = threadpool_limits(limits=cores)
    except ImportError:
        blas_cap = None

    # (1) sampling + Eloc + Jacobian: vmap(grad) + the jacrev(jacfwd) Hessian pass (var_state.py:31-32, evolutionEq.py:107-119)
    ad = {}
    for n_ad in ad_counts:
        z = O.latent_from_normals(spec, theta_cpu, torch.randn(n_ad, dim, generator=g, dtype=torch.float64))

        def autodiff():
            x, _ = vo.flow_sample(z)
            return eo(vo, x, 0.0)
        ad[n_ad] = _timed(autodiff)
    ns = np.asarray(ad_counts, dtype=np.float64)
    slope, icpt = np.polyfit(ns, [ad[n][0] for n in ad_counts], 1)
    slope_min, icpt_min = np.polyfit(ns, [ad[n][1] for n in ad_counts], 1)
    t_ad_full, t_ad_full_min = icpt + slope * n_full, icpt_min + slope_min * n_full
    t_ad_naive = ad[ad_counts[-1]][0] * n_full / ad_counts[-1]          # what scaling the largest point alone would give
    linearity = ad[ad_counts[-1]][0] / (2.0 * ad[ad_counts[-2]][0])      # 1.0 = exactly proportional to N

    # (2) tdvp.py:40-47: centred copies, F0, the two un-symmetrised Gram GEMMs (S0, SExp)
    Jg = torch.randn(n_gemm, P, generator=g, dtype=torch.float64)
    Eg = torch.randn(n_gemm, generator=g, dtype=torch.float64)
    keep = {}

    def gram_s0():
        Oc = Jg - Jg.mean(0)                           # tdvp.py:41-42
        EO = (Eg - Eg.mean())[:, None] * Oc            # tdvp.py:44
        keep["F0"], keep["EO"], keep["Oc"] = EO.mean(0), EO, Oc
        keep["S0"] = Oc.T @ Oc / n_gemm                # tdvp.py:46

    def gram_sexp():
        Ow = Eg[:, None] * keep["Oc"]
        keep["SExp"] = Ow.T @ Ow / n_gemm              # tdvp.py:47
    reps = 1 if big else 2
    t_gram1 = _timed(gram_s0, reps)
    t_gram2 = _timed(gram_sexp, reps)
    keep.pop("SExp", None)

    # (3) tdvp.py:61-64: host eigh -- at the full P (C2 / C4), at P/4 scaled by 4^3 for C3
    p_eigh = P // 4 if big else P
    S0np = np.ascontiguousarray(keep["S0"].numpy()[:p_eigh, :p_eigh])

    def eigh():
        keep["ev"], keep["V"] = np.linalg.eigh(S0np)
    t_eigh_meas = _timed(eigh, reps)
    eigh_scale = (P / p_eigh) ** 3
    t_eigh = tuple(v * eigh_scale for v in t_eigh_meas[:2])

    # (4) tdvp.py:68-70: EO @ V and its variance (C3: a random orthogonal-sized V stands in for the eigenvectors of the full matrix)
    Vt = torch.from_numpy(keep["V"]) if not big else torch.randn(P, P, generator=g, dtype=torch.float64)

    def eov():
        keep["rho"] = (keep["EO"] @ Vt).var(0)
    t_eov = _timed(eov, reps)
    if blas_cap is not None:
        blas_cap.restore_original_limits()

    scale_gemm = n_full / n_gemm
    total = t_ad_full + (t_gram1[0] + t_gram2[0] + t_eov[0]) * scale_gemm + t_eigh[0]
    total_min = t_ad_full_min + (t_gram1[1] + t_gram2[1] + t_eov[1]) * scale_gemm + t_eigh[1]
    fair = t_ad_full + t_gram1[0] * scale_gemm + t_eigh[0]     # one Gram, no SExp, no EO@V
    eigh_note = (f"eigh at the full P={P}" if not big else
                 f"eigh NOT RUN at P={P}: timed at P/4={p_eigh} ({t_eigh_meas[0]:.1f}s) and scaled by (P/{p_eigh})^3 = {eigh_scale:.0f}")
    return {"value": 1.0 / total, "unit": "TDVP steps/s", "cores": cores, "kind": "port", "workload": workload,
            "host_cpu_count": host_cpus, "affinity_count": affinity,
            "thread_sweep": {"part": f"autodiff J + Hessian at N={ad_counts[-1]} (1 warm-up + 1 timed call per setting)",
                             "seconds_by_threads": {str(t): v for t, v in sweep.items()}, "picked": cores},
            "sample": (f"reference algorithm as written on torch/numpy fp64 CPU, each part 1 warm-up + {reps} timed repetition(s) (median): "
                       f"autodiff J + Hessian at N={list(ad_counts)} -> fitted line {icpt:.2f}s + {slope * 1e3:.3f}ms/sample -> {t_ad_full:.1f}s at "
                       f"N={n_full} (scaling the largest point alone would give {t_ad_naive:.1f}s), S0 and SExp Gram GEMMs + EO@V at N={n_gemm} "
                       f"(x{int(scale_gemm)}), {eigh_note}; parts {t_ad_full:.1f}s / {(t_gram1[0] + t_gram2[0]) * scale_gemm:.1f}s / "
                       f"{t_eov[0] * scale_gemm:.1f}s / {t_eigh[0]:.1f}s -> {total:.1f}s per {n_full}-sample step (extrapolated)"),
            "seconds_per_step_extrapolated": total, "seconds_per_step_extrapolated_min": total_min,
            "parts_seconds": {**{f"autodiff_N{n}": {"median": ad[n][0], "min": ad[n][1]} for n in ad_counts},
                              f"gram_S0_N{n_gemm}": {"median": t_gram1[0], "min": t_gram1[1]},
                              f"gram_SExp_N{n_gemm}": {"median": t_gram2[0], "min": t_gram2[1]},
                              f"eigh_P{p_eigh}": {"median": t_eigh_meas[0], "min": t_eigh_meas[1]},
                              f"EO_at_V_N{n_gemm}": {"median": t_eov[0], "min": t_eov[1]}},
            "autodiff_fit": {"intercept_s": float(icpt), "slope_s_per_sample": float(slope), "full_step_s": float(t_ad_full),
                             "largest_point_scaled_s": float(t_ad_naive), "ratio_last_two_points_over_2": float(linearity)},
            "fair_variant": {"value": 1.0 / fair, "unit": "TDVP steps/s",
                             "what": "same port without the reference's redundant work (one centred Gram GEMM, no SExp, no EO@V)"},
            "note": "an eager torch.func port of the reference's algorithm, not XLA-jitted JAX-CPU (jax is not installable here): "
                    "a reported baseline, not a comparison with the reference's own runtime",
            "checked_against": "--cpu-baseline full: ONE step of the same port at the real N = 65 536 took 47.97 s on 16 threads where this "
                               "extrapolation said 46.9 s (profiles/r04_run1_bench_cpu_full.json)" if workload == "C2" else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", default="sample", choices=["sample", "full"],
                    help="sample (default): the CPU port timed piecewise on bounded samples and extrapolated (finishes in ~25 s); full: ONE step of the "
                         "port at the workload's real sample count, autodiff in 4096-sample slabs (~1 min at C2; C2 / C4 only)")
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS), help="BASELINE config (default C2, the headline)")
    ap.add_argument("--samples", type=int, default=0, help="samples per rank (default: the workload's)")
    ap.add_argument("--jacobian", default="f64", choices=["f64", "f32", "f32blocks", "i8sliced", "i8f32"],
                    help="storage type of the per-sample Jacobian (f32 = BASELINE config 5: fp32 storage, fp64 products and sums; "
                         "f32blocks: fp32 MFMA sums over blocks of 128 samples, fp64 across blocks)")
    ap.add_argument("--solver", default="cholesky", choices=["cholesky", "cholesky-dense", "matrixfree", "eigh", "eigh-dense", "eigh-matrixfree"],
                    help="cholesky: the headline (north_star's regularised on-device Cholesky solve, by the product's default route: from 4096 "
                         "parameters on a pivoted low-rank factor of S + Woodbury + refinement against the full matrix when S is rank "
                         "deficient); cholesky-dense: the blocked factorisation of the full matrix always; eigh: TDVP() exactly as the reference constructs "
                         "it (tdvp.py:21-26: spectral pseudo-inverse, diagonalShift 0, svdTol 1e-11) with the rank-revealing decomposition; "
                         "eigh-dense: the same with rocSOLVER's syevd on the full matrix -- both reported BESIDE the headline")
    ap.add_argument("--no-beside", action="store_true", help="skip the opt-in experiments measured after the timed region (`beside` in the line)")
    ap.add_argument("--overlap-allreduce", action="store_true",
                    help="N > 1: TDVP(overlapAllreduce=True) -- the S-build by groups of tile rows, every finished group all-reduced on a side stream "
                         "while the next is formed (off by default; for an A/B on a multi-GPU node: rank_stage_ms.allreduce is then the EXPOSED part)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="self-launched ranks (--gpus N > 1 without a launcher): seconds before they are stopped")
    args = ap.parse_args()

    from vmc_pde_amd import launch
    if args.gpus > 1 and not launch.under_launcher():
        # not under torch.distributed.run: be the launcher.  N fresh interpreters, one per GPU; this process makes no GPU call before or
        # after, relays rank 0's record and fails if any rank fails (mpi_wrapper.py:9-11: the reference gets its ranks from mpirun)
        try:
            rc, out = launch.run_ranks([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], args.gpus, timeout=args.launch_timeout,
                                       relay=lambda line: print(line, end="", flush=True))
        except launch.LauncherSignalled as ex:              # (SIGTERM / SIGINT to the launcher: the ranks have been stopped)
            sys.exit(128 + ex.signum)
        if rc == 0 and not any(l.lstrip().startswith("{") for l in out.splitlines()):
            print("bench: the ranks exited cleanly but rank 0 printed no record", file=sys.stderr)
            rc = 1
        sys.exit(rc)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit(f"bench: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)          # (rehearsals put several ranks on one GPU)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    backend = os.environ.get("VMCPDE_DIST_BACKEND", "nccl")     # "gloo": CPU-backend rehearsal of the multi-rank path
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    import vmc_pde_amd as V
    from vmc_pde_amd import mpi_wrapper as mpi
    if not args.samples:
        args.samples = WORKLOADS[args.workload]["n_samp"]
    experimental = args.jacobian in ("i8sliced", "i8f32") or args.solver in ("matrixfree", "eigh-matrixfree")
    vs, eq, tdvp, theta = build_problem(device, args.workload, jacobian_dtype=torch.float64 if args.jacobian in ("f64", "i8sliced") else torch.float32,
                                        experimental=experimental)
    if args.jacobian in ("i8sliced", "i8f32"):      # experiments, reported BESIDE the fp64 headline, never instead of it
        tdvp.int8Gram = True
    if args.jacobian == "f32blocks":     # fp32 accumulation inside 128-sample blocks is not exactly PSD: needs the stronger shift
        tdvp.f32BlockGram = True
        tdvp.diagonalShift = F32BLOCKS_DIAG_SHIFT
    if args.solver == "matrixfree":       # EXPERIMENT, reported beside the headline: the same regularised solve without an assembled S
        tdvp.matrixFree = True
    elif args.solver == "cholesky-dense":
        tdvp.choleskyEngine = "dense"
    elif args.solver != "cholesky":
        tdvp.solver, tdvp.diagonalShift, tdvp.svdTol = "eigh", 0.0, 1e-11
        tdvp.spectralEngine = "dense" if args.solver == "eigh-dense" else "auto"
        if args.solver == "eigh-matrixfree":
            tdvp.matrixFree = True                               # EXPERIMENT: the reference-default solve without an assembled S
    tdvp.overlapAllreduce = bool(args.overlap_allreduce)
    P = vs.numParameters
    n_total = args.samples * world
    kwargs = dict(evolutionEq=eq, psi=vs, nSamplesTDVP=n_total, nSamplesObs=n_total, timings=None, observables=False)

    stage_events = {}
    tdvp._event_hook = lambda name, a, b: stage_events.setdefault(name, []).append((a, b))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the host only enqueues (it runs ~a step ahead of the device); a generation-2 collection of the interpreter's garbage collector in
    # the middle of the timed steps (tens of ms with torch's object graph) would starve the device of launches: collect now, keep the
    # collector out of the timed region, restore it afterwards.  No work of the step is skipped.  (Before the warm-up, not between warm-up
    # and timed steps: the device must not idle there -- after 8 ms of idling the chip's power management makes the next S-build 1.8 ms
    # slower, tools/diag/gram_between_phases_probe.py.)
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
    for _ in range(args.warmup):
        tdvp(theta, 0.0, **kwargs)
    barrier()
    stage_events.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        update, _ = tdvp(theta, 0.0, **kwargs)
    t_enqueued = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    gc.unfreeze()
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert torch.isfinite(update).all(), "non-finite TDVP update"
    tdvp.check("last timed step")          # the health record of the final solve (the earlier ones were checked by the calls that followed them)
    resid = float(tdvp.solverResidual)

    # HIP-event brackets on the launch stream inside the timed steps, averaged over the steps: what this rank spent in its shard's
    # flow kernels and Gram, in the step's one collective (waiting for the slowest rank included) and in the replicated part
    # (finalize + Cholesky + substitution + diagnostics)
    mine = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in stage_events.items()}
    gram_ms = mine.get("gram", float("nan"))
    order = ["sample+flow", "gram", "allreduce", "replicated"]
    # every timed step on its own, first bracket's start to last bracket's end on the launch stream: `ms_per_step` is the mean the contract
    # asks for; median and maximum next to it show whether one stalled step (a busy host, a neighbour on the node) is inside that mean
    per_step = None
    if all(len(stage_events.get(k, [])) == args.steps for k in ("sample+flow", "replicated")):
        spans = np.asarray([stage_events["sample+flow"][i][0].elapsed_time(stage_events["replicated"][i][1]) for i in range(args.steps)])
        per_step = {"median": float(np.median(spans)), "min": float(spans.min()), "max": float(spans.max()), "slowest_step": int(spans.argmax()),
                    "first": float(spans[0]),
                    "n_over_1p25_median": int((spans > 1.25 * np.median(spans)).sum()),
                    "host_enqueue_ms_per_step": t_enqueued / args.steps * 1e3,
                    "what": "HIP-event span of each timed step on the launch stream (start of sampling to end of the solve's diagnostics); "
                            "host_enqueue = wall time the host needed to enqueue a step (it must stay below ms_per_step to keep the device fed)"}
    per_rank = [[mine.get(k, float("nan")) for k in order]]
    if world > 1:
        t_st = torch.tensor(per_rank[0], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        gathered = [torch.empty_like(t_st) for _ in range(world)]
        dist.all_gather(gathered, t_st)
        per_rank = [[float(v) for v in g.cpu()] for g in gathered]
    from vmc_pde_amd import _native as nat
    npk = nat.lib().vmcpde_gram_packed_doubles(P)
    payload_bytes = int((npk + 2 * P + 4) * 8)
    rccl = collective_record(device, world, rank, backend, payload_bytes) if world > 1 else None
    # the replicated solve must give every rank the same bits (deterministic kernels on identical reduced inputs)
    import hashlib
    upd_host = update.detach().cpu().numpy()
    digests = [hashlib.sha256(upd_host.tobytes()).hexdigest()]
    if world > 1:
        digests = [None] * world
        dist.all_gather_object(digests, hashlib.sha256(upd_host.tobytes()).hexdigest())

    # outside the timed region: per-stage wall times with a device sync at each section boundary (the reference's
    # timing sections, tdvp.py:116-128), and one Heun integrator step = 2 TDVP steps (stepper.py:135-137)
    tdvp._event_hook = None
    ratio = float(tdvp._ratio_host[0]) if tdvp._ratio_host is not None else None
    stage_ms, heun_ms = {}, None
    if world == 1:
        tm = V.util.Timings()
        for _ in range(5):
            tdvp(theta, 0.0, **{**kwargs, "timings": tm})
        stage_ms = {k: {"median": float(np.median(v)) * 1e3, "min": float(np.min(v)) * 1e3} for k, v in tm.timing_dict.items()}
        # (a tiny step: the timing does not depend on it, and the random-init C3 network is so stiff that its Jacobian grows from 4e7 to
        # 4e19 over a predictor step of 1e-4 -- the steppers now refuse a failed solve, tools/diag/c3_i8_debug.py)
        stepper = V.FixedStepper(timeStep=1e-8, mode="Heun", maxStep=1e-8, increase_fac=1.0)
        ts = []
        try:
            for _ in range(3):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                stepper.step(0.0, tdvp, theta, **kwargs)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t1)
            heun_ms = float(np.median(ts)) * 1e3
        except FloatingPointError as exc:      # reported, not fatal: the headline figures above are complete
            heun_ms = None
            print(f"bench: Heun probe step refused ({exc})", file=sys.stderr)
    flops = float(args.samples) * P * (P + 1)
    achieved = flops / (gram_ms * 1e-3) / 1e12
    peak_meas = measure_mfma_peak(device)

    if rank == 0:
        peak_spec = PEAK_F32_MFMA_TFLOPS if args.jacobian == "f32blocks" else PEAK_F64_MFMA_TFLOPS
        i8_pairs = {"i8sliced": 21.0, "i8f32": 10.0}.get(args.jacobian)
        if i8_pairs:      # the roofline of these variants is the int8 matrix pipe: 21 (10) digit-pair products per fp64 product
            achieved = i8_pairs * flops / (gram_ms * 1e-3) / 1e12
            peak_spec = PEAK_I8_MFMA_POPS * 1e3
        # HBM traffic of the Gram kernel: NOT measured in this run (PMC counters need their own rocprofv3 passes); the figure of
        # the committed counter profile of the same workload is quoted, with its source
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, "profiles", "gram_pmc_traffic.json")
        if os.path.exists(pmc) and args.workload == "C2" and args.jacobian == "f64" and args.samples == N_SAMP:
            try:
                rec = json.load(open(pmc))
                traffic = rec.get("hbm_bytes_per_launch")
                traffic_source = f"profiles/gram_pmc_traffic.json ({rec.get('source', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes')}; {rec.get('date', 'round 1')}) -- static file, not this run"
            except Exception:
                traffic = None
        steps_per_s = args.steps / elapsed
        line = {
            "metric": "TDVP steps/sec at N_samp=65k, N_param=4k" if args.workload == "C2" else f"TDVP steps/sec, workload {args.workload}",
            "value": steps_per_s * (n_total / WORKLOADS[args.workload]["n_samp"]),
            "unit": f"TDVP steps/s ({WORKLOADS[args.workload]['n_samp']}-sample steps)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_LABEL[args.jacobian], "data": "synthetic",
            "config": {"workload": WORKLOADS[args.workload]["desc"],
                       "n_param": P, "n_samp_per_gpu": args.samples, "n_samp_total": n_total,
                       "solver": "cholesky" if args.solver.startswith("cholesky") else ("cholesky, matrix-free (EXPERIMENT: S never assembled)" if args.solver == "matrixfree" else "eigh (TDVP() with the reference's constructor defaults: spectral pseudo-inverse)"),
                       **({"cholesky_engine": tdvp.choleskyInfo} if (args.solver.startswith("cholesky") or args.solver == "matrixfree") else {"spectral": tdvp.spectralInfo}),
                       "gram_operands": OPERAND_LABEL[args.jacobian] if tdvp._uncentred_ok else "centred in staging",
                       "mean2_over_var_max": ratio,
                       "diagonal_shift": float(tdvp.diagonalShift), "svd_tol": float(tdvp.svdTol), "solver_residual": resid,
                       "parallelism": f"dp{world} (samples sharded, one all-reduce of (S,F) per step)"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak_spec, "unit": "TFLOP/s" if not i8_pairs else "TOP/s (int8)",
                         "frac": achieved / peak_spec, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": KERNEL_LABEL[args.jacobian], "ms_per_launch": gram_ms,
                         "algorithmic_flops_per_launch": flops * (i8_pairs or 1.0),
                         "fp64_equivalent_tflops": flops / (gram_ms * 1e-3) / 1e12, "measured_mfma_f64_peak": peak_meas,
                         "frac_of_measured_peak": achieved / peak_meas if (peak_meas > 0 and not i8_pairs) else None},
            "stages_ms": stage_ms, "step_span_ms": per_step, "heun_step_ms": heun_ms,
            # per-rank HIP-event averages over the timed steps (ms): the Amdahl terms of the multi-GPU step
            "rank_stage_ms": [dict(zip(order, r)) for r in per_rank],
            "allreduce_payload_bytes": payload_bytes,
            # the last timed step's update: its norm and leading entries (a run with the same total sample count on another number of
            # ranks draws the same global batch, so these agree to the sums' round-off), and whether every rank holds the same bits
            "update": {"norm": float(np.linalg.norm(upd_host)), "head": [float(v) for v in upd_host[:8]], "sha256": digests[0],
                       "identical_on_all_ranks": len(set(digests)) == 1},
        }
        if world > 1:
            # the Amdahl terms read off this run: what one GPU would need for the same n_total samples (its own shard's stages x N, the
            # replicated part once) against this step -- derived from the per-rank HIP-event brackets above, not a separate run
            sf, gr, ar, rep = (float(np.max([r[i] for r in per_rank])) for i in range(4))
            line["rccl"] = rccl
            # the collective inside the timed steps: its bracket on the launch stream is what the step WAITS for -- the whole all-reduce
            # when it follows the S-build, only its exposed tail when it is interleaved with it (--overlap-allreduce)
            line["allreduce_in_step"] = dict(tdvp.allreduceInfo or {}, exposed_ms_max_rank=ar, alone_ms=rccl["allreduce_alone_ms"])
            line["one_gpu_same_samples"] = {"kind": "modelled", "ms_per_step_from_stages": world * (sf + gr) + rep,
                                            "this_run_ms_per_step": elapsed / args.steps * 1e3, "allreduce_bracket_ms_max_rank": ar,
                                            "note": "MODELLED, not measured: N x (sample+flow + gram) + replicated of the slowest rank's brackets; the measured N = 1 "
                                                    "line of the same workload is `python bench.py --workload " + args.workload + "`"}
            # the committed N = 1 record of this workload (another box, another day): the figure a reader of this line divides by (the driver computes efficiencies itself from its own N = 1 run)
            rec = os.path.join(ROOT, "profiles", "r05_run1_bench.json" if args.workload == "C2" else f"r05_run1_bench_{args.workload}.json")
            if os.path.exists(rec) and args.jacobian == "f64" and args.solver == "cholesky":
                try:
                    one = json.loads(next(l for l in open(rec) if l.startswith("{")))
                    if one["config"]["n_samp_per_gpu"] == args.samples:
                        line["n1_record"] = {"file": os.path.relpath(rec, ROOT), "value": one["value"], "ms_per_step": one["ms_per_step"]}
                except Exception:
                    pass
        if world == 1 and not args.no_beside and args.solver == "cholesky" and args.jacobian == "f64":
            # AFTER the timed region and not part of `value`: the opt-in experiments on the same problem and the same box, so that the
            # numbers quoted beside the headline come from this run as well
            line["beside"] = beside_headline(device, args, kwargs, theta)
        if world == 1 and not args.no_cpu_baseline and args.samples == WORKLOADS[args.workload]["n_samp"]:
            line["cpu_baseline"] = (cpu_baseline_full if args.cpu_baseline == "full" else cpu_baseline)(theta.cpu(), vs.net.inds_up, vs.net.inds_down, args.workload)
            line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
