import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vmc_pde_amd as V
ref = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "published_diffusion_d8_student_t.npz"))
dts = np.diff(ref["times"])
dim = 8
sampler = V.Sampler(key=1, dim=dim, numChains=30, name="Student_t", mcmc_info={"offset": np.zeros(dim), "bound": 0.25})
vs = V.VarState(sampler, dim, 1, 3, network_args={"intmediate": (6,), "offset": np.zeros(dim), "latentSpaceName": "Student_t", "dim": dim})
stepper = V.FixedStepper(timeStep=5e-8, mode="Heun", maxStep=1e-2, increase_fac=2.0)
tdvp = V.TDVP(diagonalizeOnDevice=False)
eq = V.EvolutionEquation(dim=dim, name="diffusion")
orig = tdvp.solve
def solve(Eloc, J, logp, psi=None):
    bad = []
    for nm, t in (("Eloc", Eloc), ("J", J), ("logp", logp)):
        if not torch.isfinite(t).all(): bad.append(nm)
    r = orig(Eloc, J, logp, psi=psi)
    if bad or not torch.isfinite(r[0]).all():
        print("  nonfinite inputs:", bad, "Eloc max", float(Eloc.abs().max()), "J max", float(J.abs().max()), "S0 finite", bool(torch.isfinite(tdvp.S0).all()),
              "F0 finite", bool(torch.isfinite(tdvp.F0).all()), "ev", float(tdvp.ev[0]), float(tdvp.ev[-1]), flush=True)
        x = psi._ws[Eloc.numel()]
        idx = torch.nonzero(~torch.isfinite(Eloc.reshape(-1)))[:3].reshape(-1)
        print("  bad sample idx", idx.tolist())
    return r
tdvp.solve = solve
t = 0.0
for n in range(400):
    stepper.dt, stepper.increase_fac, stepper.maxStep = float(dts[n]), 1.0, 1.0
    try:
        dp, dt, info = stepper.step(0, tdvp, vs.get_parameters(), evolutionEq=eq, psi=vs, nSamplesTDVP=10000, nSamplesObs=10000,
                                    normFunction=lambda v, S: v @ S @ v, timings=None, integrals=None)
    except FloatingPointError as e:
        print("step", n, "t", t, "FAILED", e); break
    vs.set_parameters(dp)
    nu = float(torch.exp(vs.params["params"]["dist_params"][0]) + 1)
    if n % 25 == 0 or n > 28:
        print(n, f"t={t:.4f} nu={nu:.3f} ref_nu={np.exp(ref['dist_params'][n])+1:.3f} entropy={float(info['entropy']):.4f} ref={ref['entropy'][n]:.4f} err={float(tdvp.tdvp_error):.2e} res={float(tdvp.solverResidual):.1e} |dp|={float((dp-vs.get_parameters()).abs().max()) if False else float(tdvp.F0.abs().max()):.2e} Ldiag={[round(float(v),3) for v in vs.params['params']['L_diag'][:3]]} evmax={float(tdvp.ev[-1]):.2e} Emax={float(info['max_grad']):.2e}", flush=True)
    t += dt
