"""Cost of the observables block (tdvp.py:143-162) and of a Heun step as a user's loop runs it, at C2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import vmc_pde_amd as V

dev = torch.device("cuda", 0)
vs, eq, tdvp, theta = bench.build_problem(dev, "C2")
def run(label, steps=20, **extra):
    kw = dict(evolutionEq=eq, psi=vs, nSamplesTDVP=65536, nSamplesObs=65536, timings=None, **extra)
    for _ in range(3):
        tdvp(theta, 0.0, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        tdvp(theta, 0.0, **kw)
    torch.cuda.synchronize()
    print(f"{label:50s} {(time.perf_counter() - t0) / steps * 1e3:8.3f} ms per TDVP call", flush=True)
run("observables off", observables=False)
run("observables on (reference default)")
run("observables on, integrals=None", integrals=None)
if "--heun" in sys.argv:      # (needs a trained / near-equilibrium theta: the bench's random one blows up within a few steps)
    stepper = V.FixedStepper(timeStep=1e-6, mode="Heun", maxStep=1e-6, increase_fac=1.0)
    kw = dict(evolutionEq=eq, psi=vs, nSamplesTDVP=65536, nSamplesObs=65536, timings=None)
    th = theta.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        th, dt, info = stepper.step(0.0, tdvp, th, **kw)
        vs.set_parameters(th)
    torch.cuda.synchronize()
    print(f"Heun step, user loop with observables: {(time.perf_counter() - t0) / 3 * 1e3:.3f} ms (2 TDVP calls)")
