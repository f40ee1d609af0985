import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from test_gpu_lowrank_eigh import _graded_psd, _run_pivchol_blocked
P = 700
S = _graded_psd(P, 9.0, 4096, seed=8)
d0 = np.diag(S).max()
LT, piv, r, done, head = _run_pivchol_blocked(S, 1e-14, 600, calls=(60,))
L = LT[:, :P].T
R = S.copy()
for k in range(r):
    R -= np.outer(L[:, k], L[:, k])
    if k >= 120:
        i = int(np.argmin(np.diag(R)))
        print(f"k={k+1} pivot row {piv[k]} l_kk^2/d0 {L[piv[k],k]**2/d0:.2e}  min diag(R)/d0 {np.diag(R).min()/d0:.2e} at row {i} (is pivot: {i in set(piv[:k+1].tolist())}, L[i,k]/sqrt(d0) {L[i,k]/np.sqrt(d0):.2e}), R[piv,piv]/d0 {R[piv[k],piv[k]]/d0:.2e}")
