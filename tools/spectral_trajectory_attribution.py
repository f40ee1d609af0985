"""Round 6: who is responsible for the spectral route's early cov_xx overshoot at the headline size -- the
rank-revealing ENGINE, or the reference's REGULARISER (the pseudo-inverse of tdvp.py:82-91 with a hard / soft cut at svdTol)?

The same 64 Heun steps (the reference's schedule: dt <- min(1.3 dt, 1e-2) from 1e-4, so t ~ 0.5) of the headline network (d = 2 harmonic
Kramers problem, hidden (21,21), P = 4213) at N samples per right-hand side with IDENTICAL latent draws (normals from a seeded host
generator, injected through `vmcpde_latent_from_normals` on the product side and `latent_from_normals` on the oracle side) through
    --leg product --engine lowrank|dense|auto     vmc_pde_amd.TDVP(svdTol) with that spectral engine            (GPU)
    --leg product --solver cholesky              the regularised Cholesky route of the headline, same draws     (GPU)
    --leg product --engine auto --snr            the reference's own remedy, useSNR=True (tdvp.py:87-89)        (GPU)
    --leg oracle                                 the CPU restatement of tdvp.py:36-94 as written (host eigh)    (CPU, no GPU needed)
Each leg writes theta after every step, the sample covariance / mean / entropy at the step's second stage and its wall time to --out;
    --compare a.npz b.npz ...                    prints the observables next to the exact moments and the distance between the legs.
If the three spectral legs agree with each other to the accuracy a trajectory allows, the overshoot is the regulariser's and not the
engine's.  Reference: tdvp.py:57-94, stepper.py:109-145, main.py:157-209."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

EQ = "advection_hamiltonian_wDiss"
OFFSET = [1.0, 1.0]
SEED0 = 100000


def normals_of(k, N, dim=2):
    return torch.randn(N, dim, generator=torch.Generator().manual_seed(SEED0 + k), dtype=torch.float64)


def problem(hidden):
    from oracle import vmc_oracle as O
    from helpers import make_spec
    spec = make_spec(2, 4, hidden, latent="Gauss", coupling="no_add", offset=OFFSET, seed=1)
    th0 = O.init_parameters(spec, 1)            # the reference's initialisation (output kernels 1e-5: the flow starts as the identity)
    return spec, th0


def exact_moments(ts):
    from test_gpu_published_trajectory import _oscillator_moments
    m, c = _oscillator_moments(list(ts), T=10.0, mean0=tuple(OFFSET))
    return np.asarray(m), np.asarray(c)


def leg_oracle(args):
    from oracle import vmc_oracle as O
    torch.set_num_threads(args.threads)
    spec, th0 = problem(args.hidden)
    vo, eo = O.VarStateOracle(spec, th0), O.EvolutionEquationOracle(2, EQ)
    to = O.TDVPOracle(svdTol=args.svd_tol, useSNR=args.snr)
    calls = [0]
    rec = dict(theta=[], t=[], cov=[], mean=[], entropy=[], wall=[])
    last = {}

    def f(y, t, k):
        z = O.latent_from_normals(spec, y, normals_of(calls[0], args.samples), None)
        calls[0] += 1
        u, aux = to.rhs(y, t, vo, eo, z)
        last.update(aux)
        return u, aux

    y, t, dt, t0 = th0.clone(), 0.0, 1e-4, time.perf_counter()
    for n in range(args.steps):
        dt = O.fixed_stepper_dt(dt, 1.3, 1e-2)
        try:
            y, aux = O.heun_step(f, y, 0.0, dt)
        except Exception as exc:                                     # e.g. the latent covariance of a diverged theta is no longer positive definite
            print(f"oracle step {n + 1}: the run ends here -- {type(exc).__name__}: {str(exc)[:160]}", flush=True)
            break
        obs = to.observables(aux["x"], aux["logp"], aux["Eloc"])
        t += dt
        rec["theta"].append(y.numpy().copy()); rec["t"].append(t); rec["cov"].append(obs["covar"].numpy()); rec["mean"].append(obs["x1"].numpy())
        rec["entropy"].append(float(obs["entropy"])); rec["wall"].append(time.perf_counter() - t0)
        print(f"oracle step {n + 1}/{args.steps} t={t:.4f} cov_xx={float(obs['covar'][0, 0]):.4f} rank(1e-9)={int((to.ev / to.ev[-1] > 1e-9).sum())} "
              f"{time.perf_counter() - t0:.0f} s", flush=True)
        np.savez(args.out, **{k: np.asarray(v) for k, v in rec.items()}, leg="oracle", samples=args.samples, svd_tol=args.svd_tol)


def leg_product(args):
    import vmc_pde_amd as V
    from vmc_pde_amd import _native as nat
    from helpers import make_varstate
    spec, th0 = problem(args.hidden)
    N = args.samples

    class StageSampler:
        key = 1

        def __init__(self):
            self.k = 0

        def __call__(self, numSamples, dist_params, multipleOf=1):
            z = torch.empty((N, 2), dtype=torch.float64, device="cuda")
            nrm = normals_of(self.k, N).cuda()
            nat.call("vmcpde_latent_from_normals", dist_params["_model"].handle, nat.ptr(dist_params["_theta"]), nat.ptr(nrm), None, N, nat.ptr(z),
                     nat.stream())
            self.k += 1
            return z[None]

    vs = make_varstate(spec, th0, sampler=StageSampler())
    eq = V.EvolutionEquation(dim=2, name=EQ)
    if args.solver == "cholesky":
        tdvp = V.TDVP(diagonalShift=1e-4, svdTol=args.svd_tol, solver="cholesky")
    else:
        tdvp = V.TDVP(svdTol=args.svd_tol, spectralEngine=args.engine, useSNR=args.snr)
    stepper = V.FixedStepper(timeStep=1e-4, mode="Heun", maxStep=1e-2, increase_fac=1.3)
    rec = dict(theta=[], t=[], cov=[], mean=[], entropy=[], wall=[])
    engines = {}
    t, t0 = 0.0, time.perf_counter()
    for n in range(args.steps):
        try:
            dp, dt, info = stepper.step(0, tdvp, vs.get_parameters(), evolutionEq=eq, psi=vs, nSamplesTDVP=N, nSamplesObs=N,
                                        normFunction=lambda v, S: v @ S @ v, timings=None, integrals=None)
            vs.set_parameters(dp)
        except (FloatingPointError, RuntimeError) as exc:            # the trajectory left the representable range: keep what there is
            print(f"{args.solver}/{args.engine} step {n + 1}: the run ends here -- {type(exc).__name__}: {str(exc)[:160]}", flush=True)
            break
        t += dt
        rec["theta"].append(dp.cpu().numpy().copy()); rec["t"].append(t); rec["cov"].append(info["covar"].cpu().numpy()); rec["mean"].append(info["x1"].cpu().numpy())
        rec["entropy"].append(float(info["entropy"])); rec["wall"].append(time.perf_counter() - t0)
        if tdvp.spectralInfo is not None:
            key = tdvp.spectralInfo["engine"] + ("/rank %d" % tdvp.spectralInfo["rank"] if "rank" in tdvp.spectralInfo and n == args.steps - 1 else "")
            engines[key] = engines.get(key, 0) + 1
        if (n + 1) % 8 == 0 or n + 1 == args.steps:
            print(f"{args.solver}/{args.engine}{'/snr' if args.snr else ''} step {n + 1}/{args.steps} t={t:.4f} cov_xx={float(info['covar'][0, 0]):.4f} "
                  f"{time.perf_counter() - t0:.1f} s", flush=True)
    print("engines of the last right-hand side of each step:", engines, flush=True)
    np.savez(args.out, **{k: np.asarray(v) for k, v in rec.items()}, leg=f"product:{args.solver}:{args.engine}{':snr' if args.snr else ''}",
             samples=N, svd_tol=args.svd_tol)


def compare(paths, hidden):
    legs = [(os.path.basename(p), np.load(p, allow_pickle=True)) for p in paths]
    n = min(len(d["t"]) for _, d in legs)
    ts = legs[0][1]["t"][:n]
    for _, d in legs:
        assert np.allclose(d["t"][:n], ts, rtol=1e-12), "the legs ran different step schedules"
    # observables are taken at the second stage of a step, i.e. at the state of time t (the Euler predictor): compare with the exact moments there
    m_ex, c_ex = exact_moments(ts)
    marks = [k for k in (15, 31, 47, 63, n - 1) if k < n]
    print(f"{'leg':44s} " + " ".join(f"cov_xx(t={ts[k]:.2f})" for k in marks) + "   entropy(end)")
    print(f"{'exact (damped oscillator, T = 10)':44s} " + " ".join(f"{c_ex[k][0, 0]:14.4f}" for k in marks))
    for name, d in legs:
        print(f"{str(d['leg']) + ' N=' + str(int(d['samples'])):44s} " + " ".join(f"{d['cov'][k][0, 0]:14.4f}" for k in marks) + f"   {d['entropy'][n - 1]:.4f}")
    print("\ndistance between legs, |theta_a - theta_b| / |theta_a - theta_0| after k steps (same draws; 0 = the same trajectory):")
    _, th0 = problem(hidden)
    assert th0.numel() == legs[0][1]["theta"].shape[1], "pass the legs' --hidden to --compare"
    for i in range(len(legs)):
        for j in range(i + 1, len(legs)):
            a, b = legs[i][1]["theta"], legs[j][1]["theta"]
            out = []
            for k in marks:
                ref = np.linalg.norm(a[k] - th0.numpy())
                out.append(np.linalg.norm(a[k] - b[k]) / max(ref, 1e-300))
            dc = max(abs(legs[i][1]["cov"][k][0, 0] - legs[j][1]["cov"][k][0, 0]) for k in range(n))
            print(f"  {str(legs[i][1]['leg']):32s} vs {str(legs[j][1]['leg']):32s} " + " ".join(f"{v:9.2e}" for v in out) + f"   max |d cov_xx| over the run {dc:.2e}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--leg", choices=["product", "oracle"])
    ap.add_argument("--engine", default="auto", choices=["auto", "lowrank", "dense"])
    ap.add_argument("--solver", default="eigh", choices=["eigh", "cholesky"])
    ap.add_argument("--snr", action="store_true")
    ap.add_argument("--samples", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--svd-tol", type=float, default=1e-6)
    ap.add_argument("--hidden", type=lambda s: tuple(int(v) for v in s.split(",")), default=(21, 21))
    ap.add_argument("--threads", type=int, default=min(16, os.cpu_count() or 1))
    ap.add_argument("--out", default="gpurun_out/attribution_leg.npz")
    ap.add_argument("--compare", nargs="+")
    a = ap.parse_args()
    if a.compare:
        compare(a.compare, a.hidden)
    elif a.leg == "oracle":
        leg_oracle(a)
    else:
        leg_product(a)
