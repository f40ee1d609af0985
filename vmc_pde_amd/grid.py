"""Regular grid for the 2-D normalisation check (reference: vmc_fluids/grid.py; used by main.py:100-105)."""
import numpy as np


class Grid:
    def __init__(self, bounds, n_gridpoints, sym=True):   # grid.py:9-28
        bounds = np.asarray(bounds, dtype=np.float64)
        self.sym, self.dim, self.bounds, self.n_gridpoints = sym, bounds.shape[0], bounds, n_gridpoints
        self.widths = (2 if sym else 1) * bounds / n_gridpoints
        self.bin_area = float(np.prod(self.widths))
        lo = -bounds if sym else np.zeros_like(bounds)
        self.range = [[l, b] for l, b in zip(lo, bounds)]
        self.vals = [np.arange(l, b, w) for l, b, w in zip(lo, bounds, self.widths)]
        self.meshgrid = np.meshgrid(*self.vals)
        self.coords = np.moveaxis(np.array(self.meshgrid), 0, -1).reshape(n_gridpoints ** self.dim, self.dim)
