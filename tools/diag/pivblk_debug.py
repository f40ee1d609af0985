"""Where does the blocked pivoted Cholesky lose accuracy?  Progressive remainder S - L[:, :k] L[:, :k]^T of the 700 x 700 test matrix."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from test_gpu_lowrank_eigh import _graded_psd, _run_pivchol_blocked, _run_pivchol
P = 700
S = _graded_psd(P, 9.0, 4096, seed=8)
d0 = np.diag(S).max()
for name, (LT, piv, r, done, head) in (("blocked", _run_pivchol_blocked(S, 1e-15, 600, calls=(60,))), ("unblocked", _run_pivchol(S, 1e-15, 600, chunks=(600,)))):
    L = LT[:, :P].T
    print(name, "rank", r, "done", done, "head", head / d0)
    R = S.copy()
    worst = []
    for k in range(r):
        R -= np.outer(L[:, k], L[:, k])
        dmin = np.diag(R).min()
        worst.append(dmin)
        if k % 16 == 15 or k == r - 1:
            print(f"  k={k+1:4d} pivot row {piv[k]:4d} l_kk^2/d0 {L[piv[k], k]**2/d0:.2e} min diag(R)/d0 {dmin/d0:.2e} max|R|/d0 {np.abs(R).max()/d0:.2e}")
    # the same pivot ORDER in extended precision: how far are the columns from the exact factor with that order?
    Sl = S.astype(np.longdouble); Lr = np.zeros((P, r), dtype=np.longdouble)
    for k in range(r):
        p = piv[k]
        col = Sl[:, p] - Lr[:, :k] @ Lr[p, :k]
        col[piv[:k]] = 0
        Lr[:, k] = col / np.sqrt(col[p])
    err = np.abs(L - Lr.astype(np.float64)).max(axis=0) / np.sqrt(d0)
    rel = np.linalg.norm(L - Lr.astype(np.float64), axis=0) / np.linalg.norm(Lr.astype(np.float64), axis=0)
    for k in list(range(0, r, 16)) + [r - 2, r - 1]:
        print(f"  col {k:4d}: pivot^2/d0 {float(Lr[piv[k], k]**2)/d0:.2e} max|dL|/sqrt(d0) {err[k]:.2e} rel {rel[k]:.2e}")
    Rl = Sl - Lr @ Lr.T
    print("  exact-order remainder: min diag / d0", float(np.diag(Rl).min()) / d0, "max", float(np.abs(Rl).max()) / d0)
