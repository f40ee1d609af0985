#!/usr/bin/env python3
"""The reference driver loop (vmc_fluids/main.py:60-210, plumbing only: no plots) on vmc_pde_amd.

    python examples/main.py [--mode mwe|diffusion|harmonicOsc|harmonicOsc_diff] [--t-end 0.05] [--samples 10000] [--gpus N]

`--gpus N` (N > 1): one process per GPU (the reference: `mpirun -n N python main.py`, mpi_wrapper.py:9-11) -- this process starts the N ranks
itself (vmc_pde_amd/launch.py) unless a launcher already did (torchrun: RANK / WORLD_SIZE in the environment); the samples are sharded, the
packed (S, F) buffer is all-reduced once per right-hand side over RCCL, rank 0 prints and writes the records.

Same objects, same call sequence, same printed diagnostics and `infos` keys as the reference; arrays are fp64 torch tensors
on the GPU.  The `infos` record is written to <wdir>/infos.npz (`--infos infos.hdf5`: the reference's HDF5 file, where h5py exists), and -- what the reference lacks -- the parameters are
checkpointed every step to <wdir>/checkpoint.pt so a run can be resumed with --resume.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vmc_pde_amd import var_state, sampler as sampler_mod, evolutionEq as evolutionEq_mod, tdvp, stepper, util, grid as grid_mod  # noqa: E402


def norm_fun(v, S):          # main.py:24-26
    return v @ S @ v


mode_dict = {  # main.py:33-38
    "harmonicOsc": {"offset": np.ones(2), "dim": 2, "latent_space_name": "Gauss", "mcmcbound": 0.25, "gridbound": 8., "symgrid": True, "evolution_type": "advection_hamiltonian"},
    "harmonicOsc_diff": {"offset": np.array([1., 0, 0, 1, 0, 0]), "dim": 6, "latent_space_name": "Gauss", "mcmcbound": 0.25, "gridbound": 8., "symgrid": True, "evolution_type": "advection_hamiltonian_wDiss"},
    "diffusion": {"offset": np.zeros(8), "dim": 8, "latent_space_name": "Student_t", "mcmcbound": 0.25, "gridbound": 10., "symgrid": True, "evolution_type": "diffusion"},
    "diffusion_anisotropic": {"offset": np.zeros(12), "dim": 12, "latent_space_name": "Gauss", "mcmcbound": 0.25, "gridbound": 10., "symgrid": True, "evolution_type": "diffusion_anisotropic"},
    "mwe": {"offset": np.zeros(2), "dim": 2, "latent_space_name": "Gauss", "mcmcbound": 0.25, "gridbound": 10., "symgrid": True, "evolution_type": "diffusion"},
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="mwe", choices=sorted(mode_dict))
    ap.add_argument("--t-end", type=float, default=5.0)
    ap.add_argument("--samples", type=int, default=10000)
    ap.add_argument("--wdir", default=None)
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--hidden", type=int, nargs="*", default=None, help="widths of the conditioner MLPs (default: the reference's (dim//2,))")
    ap.add_argument("--diagonal-shift", type=float, default=0.0, help="tdvp.py:50-51: S += shift * diag(S)")
    ap.add_argument("--solver", default="auto", choices=["auto", "eigh", "cholesky"],
                    help="auto/eigh: the reference's spectral pseudo-inverse; cholesky: on-device regularised Cholesky (use with --diagonal-shift)")
    ap.add_argument("--save-every", type=int, default=1, help="write the run record every this many steps (and at the end)")
    ap.add_argument("--infos", default="infos.npz", help="file name of the run record in --wdir: *.npz (numpy), or *.hdf5 = the reference's "
                                                         "infos.hdf5, one dataset per key (util.py:29-32; needs h5py)")
    ap.add_argument("--svd-tol", type=float, default=1e-11,
                    help="spectral cut-off / absolute floor; wide networks (P in the thousands) need ~1e-6 with --diagonal-shift 1e-4")
    ap.add_argument("--matrix-free", action="store_true",
                    help="EXPERIMENT: vmc_pde_amd.experimental.TDVP(matrixFree=True) -- S is never assembled (needs a rank-deficient S: large networks in low dimension)")
    ap.add_argument("--no-integrals", action="store_true",
                    help="skip the three small-ball integrals (tdvp.py:152-162; the reference evaluates them on every call and its "
                         "published records hold them, so they are on by default)")
    ap.add_argument("--gpus", type=int, default=1, help="ranks = GPUs of this node the Monte-Carlo samples are sharded over (backend nccl = RCCL; "
                                                        "VMCPDE_DIST_BACKEND=gloo rehearses it with all ranks on one GPU)")
    ap.add_argument("--launch-timeout", type=float, default=None,
                    help="--gpus N > 1 without a launcher: stop the ranks after this many seconds (default: no limit -- a run to --t-end may take "
                         "hours; a rank that dies, or a signal to this process, stops the others either way)")
    args = ap.parse_args()
    from vmc_pde_amd import launch
    if args.gpus > 1 and not launch.under_launcher():     # be the launcher: N fresh rank processes; this one never touches the GPU
        try:
            rc, _ = launch.run_ranks([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], args.gpus, timeout=args.launch_timeout,
                                     relay=lambda line: print(line, end="", flush=True))
        except launch.LauncherSignalled as ex:              # (the ranks have been stopped)
            rc = 128 + ex.signum
        sys.exit(rc)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
        torch.cuda.set_device(dev)
        backend = os.environ.get("VMCPDE_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    if rank != 0:                                           # one narrator: rank 0 prints and writes the records
        import builtins
        builtins.print = lambda *a, **k: None
    m = mode_dict[args.mode]
    dim, offset = m["dim"], m["offset"]
    initKey = 1

    smp = sampler_mod.Sampler(dim=dim, numChains=30, name=m["latent_space_name"], mcmc_info={"offset": offset, "bound": m["mcmcbound"]})   # main.py:69
    vState = var_state.VarState(smp, dim, initKey, 4, network_args={"intmediate": tuple(args.hidden) if args.hidden else (dim // 2,) * 1, "offset": offset,
                                                                     "latentSpaceName": m["latent_space_name"], "dim": dim})              # main.py:73
    print(f"Number of Model parameters: {vState.numParameters}")
    x_real = torch.ones(dim, dtype=torch.float64)                                                                                          # main.py:77-96
    z_latent, _ = vState.net.apply(vState.params, x_real, evaluate=False, inv=False)
    x_back, _ = vState.net.apply(vState.params, z_latent, evaluate=False, inv=True)
    print(z_latent, x_back)
    if dim == 2:                                                                                                                           # main.py:100-105
        g = grid_mod.Grid(np.ones(dim) * m["gridbound"], 200, sym=m["symgrid"])
        print("Integral value:", float(vState.integrate(g)))

    dt, maxStep = 1e-7, 1e-2                                                                                                               # main.py:108-118
    myStepper = stepper.FixedStepper(timeStep=dt, mode='Heun', maxStep=maxStep, increase_fac=1.3)
    if args.matrix_free:
        from vmc_pde_amd.experimental import TDVP as ExperimentalTDVP
        tdvpEq = ExperimentalTDVP(diagonalShift=args.diagonal_shift, svdTol=args.svd_tol, solver=args.solver, matrixFree=True)
    else:
        tdvpEq = tdvp.TDVP(diagonalShift=args.diagonal_shift, svdTol=args.svd_tol, solver=args.solver)
    timings = util.Timings()
    evolutionEq = evolutionEq_mod.EvolutionEquation(dim=dim, name=m["evolution_type"])
    nSamplesTDVP = nSamplesObs = args.samples
    wdir = args.wdir or f"output/{args.mode}/NsamplesTDVP{nSamplesTDVP}_NsamplesObs{nSamplesObs}/"
    os.makedirs(wdir, exist_ok=True)

    t = 0.0
    infos = {"times": [], "ev": [], "snr": [], "solver_res": [], "tdvp_error": [], "dist_params": []}                                     # main.py:157
    ckpt = os.path.join(wdir, "checkpoint.pt")
    if args.resume and os.path.exists(ckpt):
        st = torch.load(ckpt, weights_only=False)
        vState.set_parameters(st["theta"])
        t, myStepper.dt, smp._calls = st["t"], st["dt"], st["sampler_calls"]
        infos = st.get("infos", infos)          # the run record continues where the interrupted run stopped
        if "tdvp" in st:
            tdvpEq.load_state_dict(st["tdvp"])  # the lagged means of the moments / Gram: the resumed run repeats the uninterrupted one
        print(f"resumed at t = {t} with {len(infos['times'])} recorded steps")

    def host(v):
        return v.detach().cpu() if torch.is_tensor(v) else v
    while t < args.t_end + dt:                                                                                                             # main.py:159
        t1 = time.perf_counter()
        dp, dt, info = myStepper.step(0, tdvpEq, vState.get_parameters(), evolutionEq=evolutionEq, psi=vState, nSamplesTDVP=nSamplesTDVP,
                                      nSamplesObs=nSamplesObs, normFunction=norm_fun, timings=timings, integrals=not args.no_integrals)
        vState.set_parameters(dp)
        infos["times"].append(t)
        # the run record (the `infos` keys and shapes are what the reference's plot scripts read, main.py:163-190) ...
        for key, value in info.items():
            infos.setdefault(key, []).append(host(value))
        if tdvpEq.ev is not None:      # spectral quantities exist on the eigh path only (not with --solver cholesky)
            infos["ev"].append(host(tdvpEq.ev))
            infos["snr"].append(host(tdvpEq.snr))
        infos["solver_res"].append(host(tdvpEq.solverResidual))
        infos["tdvp_error"].append(host(tdvpEq.tdvp_error))
        infos["dist_params"].append(host(vState.params['params']['dist_params'].clone()))
        # ... and one status line per step
        sections = {k: v[-1] for k, v in timings.timing_dict.items() if v}
        route = (tdvpEq.choleskyInfo or tdvpEq.spectralInfo or {}).get("engine", "-")
        print(f"step {len(infos['times']):4d}  t = {t:.6f}  dt = {dt:.3e}  residual {float(tdvpEq.solverResidual):.2e}  tdvp_error {float(tdvpEq.tdvp_error):.3e}  "
              f"entropy {float(info['entropy']):.4f}  mean {np.round(host(info['x1']).numpy(), 3).tolist()}  solve: {route}  "
              f"[{', '.join(f'{k} {1e3 * v:.1f} ms' for k, v in sections.items())}; step {1e3 * (time.perf_counter() - t1):.1f} ms]", flush=True)
        t = t + dt
        # checkpoint (parameters, clock, sampler stream position AND the run record) -- written next to the target and renamed,
        # so a kill in the middle of the write leaves the previous checkpoint intact
        tdvpEq.check("before the checkpoint")      # a failed solve must not reach the checkpoint (the stepper checked already)
        state = tdvpEq.state_dict()                # (every rank: it waits for this rank's asynchronous records)
        if rank == 0:
            tmp = ckpt + ".tmp"
            torch.save({"theta": vState.get_parameters().cpu(), "t": t, "dt": myStepper.dt, "sampler_calls": smp._calls, "infos": infos,
                        "tdvp": state}, tmp)
            os.replace(tmp, ckpt)
            if len(infos["times"]) % args.save_every == 0:
                util.store_infos(wdir, infos, args.infos)
    if rank == 0:
        print("wrote", util.store_infos(wdir, infos, args.infos))                                                                          # main.py:209
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
