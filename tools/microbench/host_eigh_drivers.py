"""Host eigensolver for the spectral solve at the reference's sizes (tdvp.py:61-64 uses np.linalg.eigh): numpy against scipy's LAPACK
drivers and thread caps, on a rank-deficient Gram matrix like the real S."""
import time, numpy as np, scipy.linalg as sl
from threadpoolctl import threadpool_limits
rng = np.random.default_rng(0)
for n in (392, 411, 1024):
    X = rng.standard_normal((n, int(0.4 * n))); S = X @ X.T / X.shape[1]
    for thr in (1, 2, 4, 8):
        with threadpool_limits(limits=thr):
            res = []
            for name, f in (("np.linalg.eigh", lambda: np.linalg.eigh(S)), ("scipy evd", lambda: sl.eigh(S, driver="evd", check_finite=False)),
                            ("scipy evr", lambda: sl.eigh(S, driver="evr", check_finite=False)), ("scipy ev", lambda: sl.eigh(S, driver="ev", check_finite=False))):
                f(); t = time.perf_counter()
                for _ in range(5): f()
                res.append(f"{name} {(time.perf_counter() - t) / 5 * 1e3:7.2f} ms")
            print(f"n={n} threads={thr}: " + " | ".join(res), flush=True)

import torch
print("torch CPU eigh:", torch.__config__.parallel_info().splitlines()[0] if hasattr(torch.__config__, "parallel_info") else "")
for n in (392, 411, 1024):
    X = rng.standard_normal((n, int(0.4 * n))); S = torch.from_numpy(X @ X.T / X.shape[1])
    for thr in (1, 4, 8, 16):
        torch.set_num_threads(thr)
        torch.linalg.eigh(S); t = time.perf_counter()
        for _ in range(5): torch.linalg.eigh(S)
        print(f"n={n} torch threads={thr}: {(time.perf_counter() - t) / 5 * 1e3:7.2f} ms", flush=True)
