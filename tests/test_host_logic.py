"""CPU tests of the host logic: sample sharding, the packed-buffer algebra behind the single all-reduce, the steppers'
control flow, and the world_size-2 path over gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import vmc_oracle as O


def test_distribute_sampling_and_first_sample_id_match_oracle():
    from vmc_pde_amd import mpi_wrapper as mpi
    assert mpi.distribute_sampling(10000) == 10000 and mpi.globNumSamples == 10000
    assert mpi.distribute_sampling(10000, localDevices=1, numChainsPerDevice=30) == O.distribute_sampling(10000, 1, 0, 1, 30)[0]
    assert mpi.globNumSamples == O.distribute_sampling(10000, 1, 0, 1, 30)[1]
    assert mpi.first_sample_id(777) == 0
    assert mpi.rank == 0 and mpi.commSize == 1


class _ToyRHS:
    """f(y, t, **rhsArgs, intStep=k) -> (k, info): linear ODE with a recorded call log (stepper.py:135-137 protocol)."""

    def __init__(self, A):
        self.A, self.calls = A, []
        self.SExp = torch.eye(A.shape[0], dtype=torch.float64)

    def __call__(self, y, t, intStep=0, **kw):
        self.calls.append((intStep, float(t)))
        return self.A @ y, {"intStep": intStep}


def test_fixed_stepper_matches_oracle_and_clones():
    from vmc_pde_amd.stepper import FixedStepper
    A = torch.tensor([[0.0, 1.0], [-1.0, -0.1]], dtype=torch.float64)
    y0 = torch.tensor([1.0, 0.5], dtype=torch.float64)
    for mode, ref in (("Heun", O.heun_step), ("Euler", O.euler_step)):
        f = _ToyRHS(A)
        st = FixedStepper(timeStep=1e-2, maxStep=5e-2, increase_fac=1.3, mode=mode)
        keep = y0.clone()
        y1, dt, info = st.step(0.25, f, y0, foo=1)
        assert dt == O.fixed_stepper_dt(1e-2, 1.3, 5e-2) and st.dt == dt
        y_ref, _ = ref(lambda y, t, k: (A @ y, {}), y0, 0.25, dt)
        assert torch.allclose(y1, y_ref, atol=1e-15) and torch.equal(y0, keep)
        assert [c[0] for c in f.calls] == ([0, 1] if mode == "Heun" else [0])
        assert info["intStep"] == (1 if mode == "Heun" else 0)
    # geometric growth saturates at maxStep (stepper.py:131)
    st = FixedStepper(timeStep=1e-7, maxStep=1e-2, increase_fac=1.3)
    for _ in range(60):
        st.step(0, _ToyRHS(A), y0)
    assert st.dt == 1e-2


def test_adaptive_heun_matches_oracle():
    from vmc_pde_amd.stepper import AdaptiveHeun
    A = torch.tensor([[0.0, 1.0], [-4.0, -0.3]], dtype=torch.float64)
    y0 = torch.tensor([1.0, 0.0], dtype=torch.float64)
    f = _ToyRHS(A)
    st = AdaptiveHeun(timeStep=0.2, tol=1e-6, maxStep=1.0)
    y1, real_dt, info = st.step(0.0, f, y0, normFunction=O.norm_fun)
    y_ref, real_ref, dt_ref, _ = O.adaptive_heun_step(lambda y, t, k: (A @ y, {}), y0, 0.0, 0.2, 1e-6, 1.0, O.norm_fun, lambda: f.SExp)
    assert torch.allclose(y1, y_ref, atol=1e-14) and real_dt == real_ref and st.dt == dt_ref
    assert len(f.calls) % 5 == 0 and [c[0] for c in f.calls[:5]] == [0, 1, 2, 3, 4]   # five RHS calls per attempt


def test_shifted_moment_algebra_of_the_packed_buffer():
    """What every rank contributes and how the replicated finalize recovers the exact centred (S, F) of tdvp.py:40-46:
    S = sum_r G_r(c)/N - delta delta^T, F = sum_r f_r(c,cE)/N - delta deltaE, for ANY common shift (c, cE)."""
    rng = np.random.default_rng(0)
    N, P = 3000, 17
    J = rng.standard_normal((N, P)) * 2 + 5 * rng.standard_normal(P)
    E = rng.standard_normal(N) + 3
    Oc, Ec = J - J.mean(0), E - E.mean()
    S_ref, F_ref = Oc.T @ Oc / N, (Ec[:, None] * Oc).mean(0)
    c, cE = J[:100].mean(0), E[:100].mean()          # a lagged / partial estimate of the means
    shards = [slice(0, 1000), slice(1000, 3000)]
    G = sum((J[s] - c).T @ (J[s] - c) for s in shards)
    rs = sum((J[s] - c).sum(0) for s in shards)
    rd = sum(((J[s] - c) * (E[s] - cE)[:, None]).sum(0) for s in shards)
    sE = sum(E[s].sum() for s in shards)
    delta, dE = rs / N, sE / N - cE
    assert np.allclose(G / N - np.outer(delta, delta), S_ref, rtol=1e-12, atol=1e-12)
    assert np.allclose(rd / N - delta * dE, F_ref, rtol=1e-12, atol=1e-12)
    assert np.allclose(c + delta, J.mean(0))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vmc_pde_amd import mpi_wrapper as mpi
    n = 10001
    per = mpi.distribute_sampling(n)
    first = mpi.first_sample_id(n)
    # the single packed all-reduce: every rank contributes its shard's partial sums
    rng = np.random.default_rng(5)
    X = torch.as_tensor(rng.standard_normal((n, 6)))
    mine = X[first:first + per]
    pack = torch.cat([(mine.T @ mine).reshape(-1), mine.sum(0), torch.tensor([float(per)])])
    mpi.allreduce_sum_(pack)
    ok = torch.allclose(pack[:36].reshape(6, 6), X.T @ X, atol=1e-9) and float(pack[-1]) == n
    # reference-named generic reducers (mpi_wrapper.py:129-274) on (1, n_local, d) data
    mpi.globNumSamples = n
    gm = mpi.global_mean(mine[None])
    gv = mpi.global_variance(mine[None])
    gc = mpi.global_covariance((mine - X.mean(0))[None])
    ok = ok and torch.allclose(gm, X.mean(0), atol=1e-12) and torch.allclose(gv, X.var(0, unbiased=False), atol=1e-12)
    ok = ok and torch.allclose(gc, torch.cov(X.T, correction=0), atol=1e-12)
    q.put((rank, per, first, mpi.rank, mpi.commSize, bool(ok)))
    dist.destroy_process_group()


def test_world_size_2_gloo_sharding_and_single_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, per0, first0, rk0, cs0, ok0), (r1, per1, first1, rk1, cs1, ok1) = res
    assert (per0, per1) == (5001, 5000) and (first0, first1) == (0, 5001)     # mpi_wrapper.py:83-86,113-126
    assert (rk0, rk1, cs0, cs1) == (0, 1, 2, 2) and ok0 and ok1
    for r in range(2):
        assert O.distribute_sampling(10001, 2, r)[0] == (per0, per1)[r]


def _worker8(rank, world, port, q):
    """One of eight ranks of a ragged sample count: shard sizes / offsets (mpi_wrapper.py:68-126) and the packed-buffer algebra of a
    TDVP step -- every rank contributes shifted partial sums about a COMMON lagged shift, ONE all-reduce, and the replicated
    finalize recovers the exactly centred (S0, F0, Eloc statistics) of tdvp.py:37-46."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vmc_pde_amd import mpi_wrapper as mpi
    n, P = 10003, 9                                   # 10003 = 8 * 1250 + 3: three ranks hold one sample more
    per = mpi.distribute_sampling(n)
    first = mpi.first_sample_id(n)
    rng = np.random.default_rng(11)                    # every rank generates the same global batch and takes its shard
    J = rng.standard_normal((n, P)) * 2 + 5 * rng.standard_normal(P)
    E = rng.standard_normal(n) + 3
    c, cE = J[:50].mean(0), E[:50].mean()              # the previous call's means: any common shift works
    Jm, Em = J[first:first + per] - c, E[first:first + per]
    pack = torch.as_tensor(np.concatenate([(Jm.T @ Jm).reshape(-1), Jm.sum(0), (Jm * (Em - cE)[:, None]).sum(0),
                                           [Em.sum(), (Em ** 2).sum(), np.abs(Em).sum()], [float(per)]]))
    n_payload = pack.numel()
    mpi.allreduce_sum_(pack)                           # the step's single collective
    p = pack.numpy()
    G, rs, rd = p[:P * P].reshape(P, P), p[P * P:P * P + P], p[P * P + P:P * P + 2 * P]
    sE, sE2, sAbs, cnt = p[P * P + 2 * P:]
    delta, dE = rs / cnt, sE / cnt - cE
    S0, F0 = G / cnt - np.outer(delta, delta), rd / cnt - delta * dE
    Oc, Ec = J - J.mean(0), E - E.mean()
    ok = (cnt == n and np.allclose(S0, Oc.T @ Oc / n, rtol=1e-12, atol=1e-12) and np.allclose(F0, (Ec[:, None] * Oc).mean(0), rtol=1e-11, atol=1e-12)
          and np.allclose(c + delta, J.mean(0), rtol=1e-13) and abs(sE / cnt - E.mean()) < 1e-13
          and abs(sE2 / cnt - (sE / cnt) ** 2 - E.var()) < 1e-12 and abs(sAbs / cnt - np.abs(E).mean()) < 1e-13)
    # MCMC branch: whole sweeps of localDevices * chains on every rank (mpi_wrapper.py:87-110)
    sweeps = mpi.distribute_sampling(n, localDevices=1, numChainsPerDevice=30)
    q.put((rank, per, first, sweeps, mpi.globNumSamples, n_payload, bool(ok)))
    dist.destroy_process_group()


def test_world_size_8_gloo_ragged_shards_and_packed_buffer_algebra():
    """The eight-rank layout of BASELINE config C4 (samples sharded over 8 GPUs, one all-reduce per step), rehearsed over gloo."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 8
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    pers, firsts = [r[1] for r in res], [r[2] for r in res]
    assert pers == [1251, 1251, 1251, 1250, 1250, 1250, 1250, 1250] and sum(pers) == 10003
    assert firsts == [int(v) for v in np.concatenate([[0], np.cumsum(pers)[:-1]])]
    for r in range(world):
        n_or, glob_or = O.distribute_sampling(10003, world, r)[0], O.distribute_sampling(10003, world, r, 1, 30)
        assert n_or == pers[r] and res[r][3] == glob_or[0] and res[r][4] == glob_or[1]
        assert res[r][6], f"rank {r}: packed-buffer algebra failed"
    assert res[0][5] == 9 * 9 + 2 * 9 + 4


def test_allreduce_payload_of_a_c4_step(lib_built):
    """The one collective of a step moves [Gram tiles | sum O | sum E~O | sum E, E^2, |E| | n]: 73.6 MB at
    P = 4155 / 4213 (both 33 tile rows), 1.05 GB at P = 16181 -- the sizes bench.py reports as `allreduce_payload_bytes`."""
    from vmc_pde_amd import _native as nat
    for P, nb in ((4155, 33), (4213, 33), (16181, 127)):
        npk = nat.lib().vmcpde_gram_packed_doubles(P)
        assert npk == nb * (nb + 1) // 2 * 128 * 128
        payload = (npk + 2 * P + 4) * 8
        assert payload == {4155: 73_597_904, 4213: 73_598_832, 16181: 1_065_612_144}[P]


def test_store_infos_writes_the_run_record(tmp_path):
    """util.py:29-32: one array per key, stacked over the steps.  .npz everywhere; the reference's infos.hdf5 where h5py exists, and a
    loud error -- not a silently different file -- where it is asked for and missing."""
    from vmc_pde_amd import util
    infos = {"times": [0.0, 0.1, 0.2], "entropy": [torch.tensor(1.5), torch.tensor(1.25), torch.tensor(1.0)],
             "ev": [torch.arange(4.0), torch.arange(4.0) + 1, torch.arange(4.0) + 2]}
    path = util.store_infos(str(tmp_path), infos, "infos.npz")
    rec = np.load(path)
    assert path == os.path.join(str(tmp_path), "infos.npz") and sorted(rec.files) == ["entropy", "ev", "times"]
    assert rec["ev"].shape == (3, 4) and rec["ev"][2, 3] == 5.0 and list(rec["entropy"]) == [1.5, 1.25, 1.0]
    assert not [f for f in os.listdir(str(tmp_path)) if "tmp" in f]
    try:
        import h5py
    except ImportError:
        with pytest.raises(ImportError, match="h5py"):
            util.store_infos(str(tmp_path), infos, "infos.hdf5")
        assert util.store_infos(str(tmp_path), infos).endswith("infos.npz")
        return
    path = util.store_infos(str(tmp_path), infos)
    assert path.endswith("infos.hdf5")
    with h5py.File(path, "r") as f:
        assert sorted(f.keys()) == ["entropy", "ev", "times"] and f["ev"][:].shape == (3, 4) and f["times"][2] == 0.2


def test_gram_operator_protocol_routes_the_norm_function():
    """GramOperator (vmc_pde_amd/tdvp.py): what the steppers hand to `normFunction(v, S)` in place of the matrix SExp.  The reference's
    `v @ S @ v` (main.py:24-26) and its variants must reach the one-pass quadratic form / the two-pass product, everything else the
    dense matrix -- checked on a stand-in engine that counts which of the three was used."""
    from vmc_pde_amd.tdvp import GramOperator

    class Engine:
        def __init__(self, S):
            self._S, self.used = S, []

        @property
        def SExp(self):
            self.used.append("dense")
            return self._S

        def _sexp_form(self, v):
            self.used.append("form")
            return v @ self._S @ v

        def _sexp_matvec(self, v):
            self.used.append("matvec")
            return self._S @ v

    g = torch.Generator().manual_seed(1)
    A = torch.randn(7, 7, generator=g, dtype=torch.float64)
    S = A @ A.T
    v, w = torch.randn(7, generator=g, dtype=torch.float64), torch.randn(7, generator=g, dtype=torch.float64)
    ref = float(v @ S @ v)
    for expr, expect in ((lambda op: v @ op @ v, ["form"]), (lambda op: torch.matmul(v, op) @ v, ["form"]),
                         (lambda op: torch.dot(v, op @ v), ["matvec"]), (lambda op: torch.dot(v, torch.mv(op, v)), ["matvec"]),
                         (lambda op: v @ (op @ v), ["matvec"])):
        e = Engine(S)
        assert abs(float(expr(GramOperator(e))) - ref) < 1e-12 * abs(ref) and e.used == expect, (e.used, expect)
    e = Engine(S)
    assert abs(float(v @ GramOperator(e) @ w) - float(v @ S @ w)) < 1e-12 * abs(ref) and e.used == ["matvec"]      # a different right operand
    # every other use is the dense matrix
    for expr in (lambda op: (op - S).abs().max(), lambda op: op[1, 2] - S[1, 2], lambda op: torch.linalg.norm(op) - torch.linalg.norm(S),
                 lambda op: (op.clone() - S).abs().max(), lambda op: (torch.diagonal(op) - torch.diagonal(S)).abs().max(),
                 lambda op: (A.T @ op @ A - A.T @ S @ A).abs().max()):
        e = Engine(S)
        assert float(expr(GramOperator(e))) == 0.0 and "dense" in e.used and "form" not in e.used
    assert GramOperator(Engine(S)).shape == S.shape


def test_rhovar_from_ritz_vectors_positions_and_values_on_cpu():
    """TDVP._rhoVar_from_ritz (tdvp.py:68-70 after the rank-revealing decomposition) as host logic, on CPU tensors: the r Ritz modes sit in
    V as [negative Ritz values | P - r null vectors | non-negative Ritz values]; rhoVar of a Ritz mode is the dense formula's, of a null mode
    exactly zero -- also when some Ritz values are (round-off) negative, which the GPU tests rarely produce."""
    import vmc_pde_amd as V
    from vmc_pde_amd import global_defs
    rng = np.random.default_rng(5)
    P, n, r = 23, 400, 7
    B = rng.standard_normal((P, r))
    J = (B @ rng.standard_normal((r, n))).T                      # n x P, rank r (after centring: rank <= r)
    e = rng.standard_normal(n)
    m = J.mean(0)
    O = J - m
    S = O.T @ O / n
    w_all, V_all = np.linalg.eigh(S)
    ritz = np.argsort(w_all)[P - r:]                              # the r modes that carry the spectrum
    U = V_all[:, ritz]
    wv = w_all[ritz].copy()
    wv[0], wv[1] = -1e-17, -3e-18                                 # two round-off negatives: they sort in FRONT of the null block
    order = np.argsort(wv)
    U, wv = U[:, order], wv[order]
    null = V_all[:, np.argsort(w_all)[:P - r]]
    Vfull = np.concatenate([U[:, :2], null, U[:, 2:]], axis=1)   # what _lowrank_assemble builds for nneg = 2
    Et = e - e.mean()
    F = (Et[:, None] * O).mean(0)
    VtF = Vfull.T @ F
    G = (O * (Et ** 2)[:, None]).T @ O / n
    rho_dense = np.einsum("pk,pq,qk->k", Vfull, G, Vfull) - VtF ** 2
    tp = V.TDVP()
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=global_defs.tReal)
    tp._lazy = dict(JT=T(J.T), P=P, n=n, mean=T(m), e=T(e), Emean=T(np.array(e.mean())), Ntot=float(n), owner=None, gen=None)
    tp.VtF = T(VtF)
    rho = tp._rhoVar_from_ritz(dict(U=T(U), wv=T(wv), r=r, P=P)).numpy()
    pos = np.r_[0, 1, np.arange(P - r + 2, P)]
    top = np.abs(rho_dense).max()
    assert np.abs(rho[pos] - rho_dense[pos]).max() < 1e-12 * top
    nullpos = np.setdiff1d(np.arange(P), pos)
    assert not rho[nullpos].any() and np.abs(rho_dense[nullpos]).max() < 1e-12 * top


def test_frozen_pivot_set_decision():
    """The host-side rule of the static low-rank route (round 6): which calls factor S on the previous pivot set -- a pure function of values
    every rank holds (lowrank.frozen_set_decision)."""
    from vmc_pde_amd import lowrank as LR
    f = LR.frozen_set_decision
    n = LR.FROZEN_REFRESH_AFTER
    assert f(True, 206, 0, False, None) and f(True, 206, n - 1, True, (1e-6, 0.0))
    assert not f(True, 206, n, True, (1e-6, 0.0))                       # the set is refreshed after n calls in a row
    assert not f(False, 206, 0, False, None) and not f(True, 0, 0, False, None)
    assert not f(True, 206, 2, True, (2 * LR.LOWRANK_SOLVE_RHO, 0.0))   # the remainder outgrew the refinement's contraction bound
    assert not f(True, 206, 2, True, (1e-6, 1.0))                       # a pinned pivot
    assert not f(True, 206, 2, True, (float("nan"), 0.0)) and not f(True, 206, 2, True, (float("inf"), 0.0))
    # the cadence it produces: n frozen calls, one pivoted, n frozen ...
    run, seq = 0, []
    for _ in range(2 * n + 3):
        d = f(True, 206, run, bool(seq) and seq[-1], (1e-7, 0.0))
        run = run + 1 if d else 0
        seq.append(d)
    assert seq == [True] * n + [False] + [True] * n + [False] + [True]
