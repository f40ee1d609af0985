"""Where wave 0 of a flow_eval workgroup spends its cycles (build: python -m vmc_pde_amd.build --variant fstamps -DVMC_FLOW_STAMPS;
run with VMCPDE_LIB=.../libvmcpde_fstamps.so).  Wave 0 is the critical path of a workgroup: it owns the per-sample state and
runs the coupling algebra between the cooperative MLP passes."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from vmc_pde_amd import _native as nat
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
dev = torch.device("cuda", 0)
if wl == "small":      # the reference's phase-space model: d = 6, 4 blocks x hidden (3,), different_add, P = 411, N = 10 000
    import numpy as np, vmc_pde_amd as V
    off = np.array([1.0, 0.0, 1.0, 0.0, 1.0, 0.0])
    sampler = V.Sampler(key=1, dim=6, name="Gauss", mcmc_info={"offset": off, "bound": 0.25})
    vs = V.VarState(sampler, 6, 1, 4, network_args={"intmediate": (3,), "offset": off, "latentSpaceName": "Gauss", "dim": 6, "different_add": True})
    eq = V.EvolutionEquation(dim=6, name="advection_hamiltonian_wDiss")
    N = 10000
else:
    vs, eq, tdvp, theta = bench.build_problem(dev, wl)
    N = bench.WORKLOADS[wl]["n_samp"]
x, _ = vs.sample(N)
lib = nat.lib()
out = (C.c_ulonglong * 10)()
for it in range(3):
    torch.cuda.synchronize()
    lib.vmcpde_debug_read_flow_stamps(out, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); E, J, lp = eq(vs, x, 0.0); e1.record()
    torch.cuda.synchronize()
    lib.vmcpde_debug_read_flow_stamps(out, 0)
v = [int(a) for a in out]
names = ["fwd gather+barrier", "fwd MLP layers (wave 0 busy)", "fwd coupling algebra", "fwd weight stage + barrier (+latent fwd)", "fwd layer barriers (+latent bwd)", "rev forward-store", "rev coupling algebra",
         "rev MLP backward+stores", "rest"]
tot = sum(v[:9])
print(f"{wl}: Eloc+J {e0.elapsed_time(e1):.3f} ms; {v[9]} workgroups; wave-0 cycles per workgroup {tot / v[9]:.0f} (shader clock)")
for nme, a in zip(names, v[:9]):
    print(f"  {nme:26s} {a / v[9]:10.0f} cycles  {100.0 * a / tot:5.1f} %")
