"""Quadrature grid for the normalisation check of a low-dimensional density (reference: vmc_fluids/grid.py, used by
main.py:100-105 as `vState.integrate(Grid(bounds, n, sym))`): a midpoint-free rectangle rule on n points per axis,
left-closed cells of width (upper - lower) / n, with lower = -bounds (sym) or 0."""
import numpy as np


class Grid:
    def __init__(self, bounds, n_gridpoints, sym=True):
        upper = np.atleast_1d(np.asarray(bounds, dtype=np.float64))
        lower = -upper if sym else np.zeros_like(upper)
        self.sym, self.dim, self.bounds, self.n_gridpoints = sym, upper.size, upper, n_gridpoints
        self.range = np.stack([lower, upper], axis=1).tolist()
        self.widths = (upper - lower) / n_gridpoints
        self.bin_area = float(self.widths.prod())
        # node k of axis a: lower_a + k * width_a  (k = 0 .. n-1); same point order as np.meshgrid(..., indexing="xy") raveled
        self.vals = [lo + w * np.arange(n_gridpoints) for lo, w in zip(lower, self.widths)]
        self.meshgrid = np.meshgrid(*self.vals)
        self.coords = np.stack([m.reshape(-1) for m in self.meshgrid], axis=-1)
