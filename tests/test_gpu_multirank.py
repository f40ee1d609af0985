"""The N > 1 path on the one available GPU: two ranks (gloo, both on cuda:0) shard the samples, all-reduce the packed
(S, F) buffer once and run the replicated solve; the result must equal the single-rank run on the same global sample
set (the Philox sampler indexes draws by global sample id, so the union of the shards IS the single-rank batch)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N, DIM, DEPTH, HIDDEN = 6001, 2, 3, (5, 5)     # odd N: ranks get 3001 / 3000 samples


# the two solvers, and both through the pivoted low-rank factor of S (forced: the default takes those routes from 1024 / 4096 parameters
# on) -- the second call of the Cholesky variant runs its static plan
SOLVERS = (("chol", dict(diagonalShift=1e-5, solver="cholesky")), ("eigh", dict(solver="eigh")),
           # the same step with the collective interleaved with the S-build by groups of tile rows (P = 281: three tile rows, three groups)
           ("chol-overlap", dict(diagonalShift=1e-5, solver="cholesky", overlapAllreduce=True)),
           ("chol-lowrank", dict(diagonalShift=1e-4, svdTol=1e-6, solver="cholesky", choleskyEngine="lowrank")),
           ("eigh-lowrank", dict(solver="eigh", spectralEngine="lowrank")),
           # ... with the SNR-regularised pseudo-inverse: rhoVar comes from the r Ritz vectors, its sample sums all-reduced inside the solve
           ("eigh-lowrank-snr", dict(solver="eigh", spectralEngine="lowrank", useSNR=True)),
           # the matrix-free experiment: no S at all; moments, every 32-column panel and every S0 v are summed over the ranks
           ("chol-matrixfree", dict(diagonalShift=1e-4, svdTol=1e-6, solver="cholesky", matrixFree=True)),
           # the int8 S-build experiment: its heavy-tail figure is all-reduced (max) so that every rank decides alike
           ("chol-int8", dict(diagonalShift=1e-4, svdTol=1e-6, solver="cholesky", int8Gram=True)))


def _problem():
    import vmc_pde_amd as V
    sampler = V.Sampler(key=77, dim=DIM, name="Gauss", mcmc_info={"offset": np.ones(DIM), "bound": 0.25})
    vs = V.VarState(sampler, DIM, 5, DEPTH, network_args=dict(intmediate=HIDDEN, offset=np.ones(DIM), latentSpaceName="Gauss", dim=DIM))
    vs.set_parameters(vs.net.init(5, out_scale=0.05))
    vs.theta_start = vs.get_parameters().clone()
    eq = V.EvolutionEquation(dim=DIM, name="advection_hamiltonian_wDiss")
    eq.eqParams[eq.name]["params"]["lam"] = 0.1
    return V, vs, eq


def _two_calls(V, vs, eq, solver_kw):
    if solver_kw.get("matrixFree") or solver_kw.get("int8Gram"):
        from vmc_pde_amd.experimental import TDVP as cls        # (the matrix-free experiment is a subclass of the product class)
    else:
        cls = V.TDVP
    tdvp = cls(**solver_kw)
    vs.sampler._calls = 0                     # every variant draws the same two batches (sample n of call k is a function of (key, k, n))
    th0 = vs.theta_start.clone()              # ... at the same two states
    u0, _ = tdvp(th0, 0.0, vs, eq, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)
    none = torch.zeros(1, 1, dtype=th0.dtype, device=th0.device)          # (the matrix-free variant has no S0)
    S0, F0 = (tdvp.S0.clone() if tdvp.S0 is not None else none), tdvp.F0.clone()
    th1 = th0 + 1e-3 * torch.sin(torch.arange(th0.numel(), dtype=th0.dtype, device=th0.device))   # same on every rank / run
    u1, _ = tdvp(th1, 0.1, vs, eq, nSamplesTDVP=N, nSamplesObs=N, timings=None, observables=False)   # second call: lagged centring when sharded
    return [t.cpu() for t in (u0, S0, F0, u1, tdvp.S0 if tdvp.S0 is not None else none, tdvp.F0)] + [float(tdvp.ElocMean), float(tdvp.ElocVar)]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    V, vs, eq = _problem()
    res = {}
    for name, kw in SOLVERS:
        res[name] = _two_calls(V, vs, eq, kw)
    q.put((rank, {k: [v.numpy() if torch.is_tensor(v) else v for v in vals] for k, vals in res.items()}))   # by value
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_match_single_rank():
    V, vs, eq = _problem()
    single = {name: _two_calls(V, vs, eq, kw) for name, kw in SOLVERS}
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in procs)
    got = {r: {k: [torch.from_numpy(v) if isinstance(v, np.ndarray) else v for v in vals] for k, vals in d.items()} for r, d in got.items()}
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for name, _ in SOLVERS:
        u0, S0, F0, u1, S1, F1, Em, Ev = single[name]
        for rank in (0, 1):
            r = got[rank][name]
            if S0.numel() > 1:
                assert float((r[1] - S0).abs().max() / S0.abs().max()) < 1e-12        # S of the first call (exact centring)
                assert float((r[4] - S1).abs().max() / S1.abs().max()) < 1e-11        # second call: lagged centring, one all-reduce
            assert float((r[2] - F0).abs().max() / F0.abs().max()) < 1e-11
            assert float((r[5] - F1).abs().max() / F1.abs().max()) < 1e-10
            assert abs(r[6] - Em) < 1e-11 * (1 + abs(Em)) and abs(r[7] - Ev) < 1e-10 * (1 + abs(Ev))
            if name.startswith("chol"):                                              # well conditioned: updates agree directly
                assert float(torch.linalg.norm(r[0] - u0) / torch.linalg.norm(u0)) < 1e-7
                assert float(torch.linalg.norm(r[3] - u1) / torch.linalg.norm(u1)) < 1e-7
            else:                                                                    # singular S: compare in the S-norm
                assert float(torch.linalg.norm(S0 @ (r[0] - u0)) / torch.linalg.norm(F0)) < 1e-8
        # the replicated solve is bit-identical on both ranks (same all-reduced inputs, deterministic kernels)
        assert torch.equal(got[0][name][0], got[1][name][0]) and torch.equal(got[0][name][3], got[1][name][3])
    # the overlapped collective moves the same bits as the single call: S, F and both updates, on one rank and on two
    for a, b in zip(single["chol"][:6], single["chol-overlap"][:6]):
        assert torch.equal(a, b)
    for rank in (0, 1):
        for a, b in zip(got[rank]["chol"][:6], got[rank]["chol-overlap"][:6]):
            assert torch.equal(a, b)


def test_abi_allreduce_on_a_single_rank_communicator():
    """The C-ABI collective end to end on the one GPU there is: unique id -> ncclCommInitRank (1 rank) -> vmcpde_allreduce in
    place on the current stream -> destroy.  A one-rank sum is the identity, which is exactly what makes it checkable here."""
    import ctypes as C
    from vmc_pde_amd import _native as nat
    torch.cuda.set_device(0)
    ident = (C.c_char * 128)()
    nat.call("vmcpde_comm_unique_id", C.cast(ident, C.c_void_p))
    assert any(b != 0 for b in ident.raw)
    comm = C.c_void_p()
    nat.call("vmcpde_comm_init_rank", C.byref(comm), 1, C.cast(ident, C.c_void_p), 0)
    assert comm.value
    g = torch.Generator(device="cuda").manual_seed(3)
    buf = torch.randn(9_191_000, generator=g, dtype=torch.float64, device="cuda")    # the size of the C2 payload (73.5 MB)
    keep = buf.clone()
    for _ in range(3):
        nat.call("vmcpde_allreduce", nat.ptr(buf), buf.numel(), comm, nat.stream())
    torch.cuda.synchronize()
    assert torch.equal(buf, keep)
    with pytest.raises(nat.VmcpdeError, match="null argument"):
        nat.call("vmcpde_allreduce", nat.ptr(buf), buf.numel(), None, nat.stream())
    nat.call("vmcpde_comm_destroy", comm)


def _worker_nccl(rank, world, port, collective, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      VMCPDE_COLLECTIVE=collective, HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    V, vs, eq = _problem()
    res = _two_calls(V, vs, eq, dict(diagonalShift=1e-5, solver="cholesky"))
    q.put((rank, [v.numpy() if torch.is_tensor(v) else v for v in res]))
    dist.barrier()
    from vmc_pde_amd import mpi_wrapper as mpi
    mpi.destroy_abi_communicator()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
@pytest.mark.parametrize("collective", ["torch", "abi"])
def test_two_ranks_over_rccl_match_single_rank(collective):
    """One rank per GPU, backend nccl (= RCCL over xGMI): the packed buffer is reduced on the device, either by
    torch.distributed or by the library's own vmcpde_allreduce.  Runs on the first box that has two GPUs."""
    V, vs, eq = _problem()
    u0, S0, F0, u1, S1, F1, Em, Ev = _two_calls(V, vs, eq, dict(diagonalShift=1e-5, solver="cholesky"))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_nccl, args=(r, 2, port, collective, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank in (0, 1):
        r = [torch.from_numpy(v) if isinstance(v, np.ndarray) else v for v in got[rank]]
        assert float((r[1] - S0).abs().max() / S0.abs().max()) < 1e-12 and float((r[4] - S1).abs().max() / S1.abs().max()) < 1e-11
        assert float((r[2] - F0).abs().max() / F0.abs().max()) < 1e-11 and float((r[5] - F1).abs().max() / F1.abs().max()) < 1e-10
        assert float(torch.linalg.norm(r[0] - u0) / torch.linalg.norm(u0)) < 1e-7
        assert float(torch.linalg.norm(r[3] - u1) / torch.linalg.norm(u1)) < 1e-7
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][3], got[1][3])       # replicated solve: bit-identical


def _bench_line(args, extra_env=None, timeout=900):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env.update(extra_env or {})
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]                       # ONE record on stdout, whatever the ranks print
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks_and_proves_who_took_part():
    """`python bench.py --gpus 2` outside any launcher (how the driver starts the N = 1 line): the parent spawns the two ranks
    (vmc_pde_amd/launch.py), rank 0's record is the only line on stdout, it carries the collective's own account of the job, the
    replicated update is the same bits on both ranks and equals the single-rank update on the same global batch (the sampler indexes
    draws by global sample id).  Rehearsed over gloo with both ranks on the one GPU (RCCL refuses two ranks on one device); on a
    multi-GPU node the same command runs backend nccl."""
    common = ["--steps", "2", "--warmup", "2", "--no-cpu-baseline", "--no-beside"]
    two = _bench_line(["--gpus", "2", "--samples", "2048", *common], {"VMCPDE_DIST_BACKEND": "gloo"})
    one = _bench_line(["--gpus", "1", "--samples", "4096", *common])
    assert two["n_gpus"] == 2 and two["config"]["n_samp_total"] == 4096 and one["config"]["n_samp_total"] == 4096
    r = two["rccl"]
    assert r["backend"] == "gloo" and r["world_size"] == 2 and r["collective"] == "torch"
    assert [e["rank"] for e in r["ranks"]] == [0, 1] and len({e["pid"] for e in r["ranks"]}) == 2
    assert all(len(b) >= 7 for b in r["device_bus_ids"]) and r["devices_distinct"] is False     # both ranks on the one GPU here
    assert r["onehot_sum"] == [1.0, 1.0] and r["every_rank_contributed_once"]
    assert r["allreduce_payload_bytes"] == two["allreduce_payload_bytes"] == one["allreduce_payload_bytes"]
    assert two["update"]["identical_on_all_ranks"] and len(two["rank_stage_ms"]) == 2
    u2, u1 = np.asarray(two["update"]["head"]), np.asarray(one["update"]["head"])
    assert abs(two["update"]["norm"] / one["update"]["norm"] - 1) < 1e-6 and np.linalg.norm(u2 - u1) < 1e-6 * np.linalg.norm(u1)
    assert two["one_gpu_same_samples"]["ms_per_step_from_stages"] > 0
    # weak scaling: `value` counts 65 536-sample steps over all ranks
    assert abs(two["value"] - 2 / (two["ms_per_step"] * 1e-3) * 2048 / 65536) < 1e-9 * two["value"]


def test_driver_loop_on_two_ranks_matches_one_rank(tmp_path):
    """examples/main.py --gpus 2 (the reference: `mpirun -n 2 python main.py`): the driver starts its own two ranks, shards the samples, reduces
    (S, F) once per right-hand side, rank 0 writes the records -- and the parameters after nine Heun steps are those of the one-rank run on the
    same global batches.  Over gloo with both ranks on the one GPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    out = {}
    for n in (1, 2):
        wdir = os.path.join(str(tmp_path), f"ranks{n}") + "/"
        cmd = [sys.executable, os.path.join(root, "examples", "main.py"), "--mode", "mwe", "--samples", "2048", "--t-end", "3e-6", "--wdir", wdir,
               "--gpus", str(n)]
        res = subprocess.run(cmd, env=dict(env, VMCPDE_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        assert res.stdout.count("Number of Model parameters: 37") == 1          # one narrator
        out[n] = (torch.load(os.path.join(wdir, "checkpoint.pt"), weights_only=False), np.load(os.path.join(wdir, "infos.npz")), res.stdout)
    (c1, r1, _), (c2, r2, _) = out[1], out[2]
    assert c1["t"] == c2["t"] and c1["sampler_calls"] == c2["sampler_calls"] and len(r1["times"]) == len(r2["times"]) >= 8
    import vmc_pde_amd as V
    th0 = None
    d12 = float(torch.linalg.norm(c2["theta"] - c1["theta"]))
    # the distance travelled: from the initial parameters (the same construction as the driver's) to the end of the one-rank run
    smp = V.Sampler(dim=2, numChains=30, name="Gauss", mcmc_info={"offset": np.zeros(2), "bound": 0.25})
    vs = V.VarState(smp, 2, 1, 4, network_args={"intmediate": (1,), "offset": np.zeros(2), "latentSpaceName": "Gauss", "dim": 2})
    th0 = vs.get_parameters().cpu()
    travelled = float(torch.linalg.norm(c1["theta"] - th0))
    assert travelled > 0 and d12 < 1e-5 * travelled, (d12, travelled)
    # the two-rank record is rank 0's shard (1024 of the 2048 samples): same law, different estimate -- and the spectrum is the global S's
    assert np.allclose(r1["ev"][-1], r2["ev"][-1], rtol=1e-9, atol=1e-12 * r1["ev"][-1].max())
    assert abs(r1["entropy"][-1] - r2["entropy"][-1]) < 0.2
