#!/opt/conda/bin/python3.9
"""Extracts small known-answer fixtures from the reference's PUBLISHED result files (HDF5 data, read as data with h5py
from /opt/conda/bin/python3.9 -- the reference code itself is not imported or executed).

    /opt/conda/bin/python3.9 tests/golden/make_published_fixtures.py

Sources (SURVEY.md section 6 / BASELINE.md section 1):
  vmc_fluids/paper_plot/data_diffusion/dim8_Gauss_NsamplesTDVP10000_NsamplesObs10000/infos.hdf5
  vmc_fluids/paper_plot/data_diffusion/dim8_StudentT_nu2_NsamplesTDVP10000_NsamplesObs10000/infos.hdf5
  vmc_fluids/paper_plot/data_phaseSpace/Wiener/Nsamples10000_T10.0/infos.hdf5      (exact_dyn.py Langevin ensemble, T = 10)
  vmc_fluids/paper_plot/data_phaseSpace/INN/NsamplesTDVP10000_NsamplesObs10000_Tdifferent/infos.hdf5   (step schedule, diagnostics)
"""
import os

import h5py
import numpy as np

REF = "/root/reference/vmc_fluids/paper_plot/"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    with h5py.File(REF + "data_diffusion/dim8_Gauss_NsamplesTDVP10000_NsamplesObs10000/infos.hdf5", "r") as f:
        g = dict(times=f["times"][:], entropy=f["entropy"][:], covar_diag=np.diagonal(f["covar"][:], axis1=1, axis2=2).copy(),
                 tdvp_error=f["tdvp_error"][:], solver_res=f["solver_res"][:], ev_first=f["ev"][0], ev_last=f["ev"][-1],
                 x1_last=f["x1"][-1], snr_first=f["snr"][0], x1=f["x1"][:], ev_step20=f["ev"][20], snr_step20=f["snr"][20],
                 # tdvp.py:143-162: the small-ball integrals, the central moments 3..6 and the largest Eloc of every step
                 integral_1sigma=f["integral_1sigma"][:], integral_05sigma=f["integral_0.5sigma"][:], integral_01sigma=f["integral_0.1sigma"][:],
                 x3=f["x3"][:], x4=f["x4"][:], x5=f["x5"][:], x6=f["x6"][:], max_grad=f["max_grad"][:])
    np.savez_compressed(os.path.join(OUT, "published_diffusion_d8_gauss.npz"), **g)
    with h5py.File(REF + "data_diffusion/dim8_StudentT_nu2_NsamplesTDVP10000_NsamplesObs10000/infos.hdf5", "r") as f:
        s = dict(times=f["times"][:], entropy=f["entropy"][:], dist_params=f["dist_params"][:, 0], tdvp_error=f["tdvp_error"][:],
                 solver_res=f["solver_res"][:], covar_diag=np.diagonal(f["covar"][:], axis1=1, axis2=2).copy(),
                 ev_first=f["ev"][0], ev_last=f["ev"][-1], snr_first=f["snr"][0],
                 integral_1sigma=f["integral_1sigma"][:], integral_05sigma=f["integral_0.5sigma"][:], integral_01sigma=f["integral_0.1sigma"][:],
                 max_grad=f["max_grad"][:])
    np.savez_compressed(os.path.join(OUT, "published_diffusion_d8_student_t.npz"), **s)
    # phase space (3 damped oscillators, evolutionEq.py:107-119): the reference's own Langevin ensemble for T = 10 up to t = 5,
    # every 10th record; and the flow run's step schedule / solver diagnostics (its temperatures differ per oscillator and
    # are not recorded, so its moments are not used)
    with h5py.File(REF + "data_phaseSpace/Wiener/Nsamples10000_T10.0/infos.hdf5", "r") as f:
        sel = slice(0, 501, 10)
        w = dict(times=f["times"][sel], x1=f["x1"][sel], covar=f["covar"][sel],
                 # fraction of the 10 000 walkers inside the three balls (exact_dyn.py), all 1201 records up to t = 12
                 wiener_times_all=f["times"][:], wiener_integral_1sigma=f["integral_1sigma"][:],
                 wiener_integral_05sigma=f["integral_0.5sigma"][:], wiener_integral_01sigma=f["integral_0.1sigma"][:])
    with h5py.File(REF + "data_phaseSpace/INN/NsamplesTDVP10000_NsamplesObs10000_Tdifferent/infos.hdf5", "r") as f:
        w.update(inn_times=f["times"][:], inn_tdvp_error=f["tdvp_error"][:], inn_solver_res=f["solver_res"][:],
                 inn_n_param=np.int64(f["ev"].shape[1]), inn_ev_first=f["ev"][0], inn_ev_last=f["ev"][-1],
                 inn_integral_1sigma=f["integral_1sigma"][:], inn_integral_05sigma=f["integral_0.5sigma"][:],
                 inn_integral_01sigma=f["integral_0.1sigma"][:], inn_x1=f["x1"][:], inn_covar_diag=np.diagonal(f["covar"][:], axis1=1, axis2=2).copy(),
                 inn_max_grad=f["max_grad"][:], inn_entropy=f["entropy"][:])
    np.savez_compressed(os.path.join(OUT, "published_phase_space.npz"), **w)
    print("phase space: wiener", w["times"].shape, "covar diag at t=5", np.diagonal(w["covar"][-1]), "inn steps", w["inn_times"].shape,
          "tdvp_error median/max", np.median(w["inn_tdvp_error"]), w["inn_tdvp_error"].max())
    for k, v in g.items():
        print("gauss", k, v.shape)
    dt = np.diff(g["times"])
    print("dt[:8]", dt[:8], "ratio", dt[1:8] / dt[:7], "max dt", dt.max(), "t_end", g["times"][-1])
    print("entropy dev", np.abs(g["entropy"] - 0.5 * 8 * np.log(2 * np.pi * np.e * (1 + 2 * g["times"]))).max())
    print("nu", np.exp(s["dist_params"][[0, -1]]) + 1)


if __name__ == "__main__":
    main()
