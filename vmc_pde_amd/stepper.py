"""ODE integrators for the parameter flow (reference: vmc_fluids/stepper.py).

Same classes, constructor arguments and `step(t, f, y, normFunction, **rhsArgs) -> (y_new, dt, info)` protocol.
The reference relies on the value semantics of immutable jnp arrays (`y += dt*k0` rebinds, stepper.py:136); with
torch tensors every update below builds a new tensor instead of mutating the caller's.
"""
import torch


def _default_norm(v, *_):
    return torch.linalg.norm(v)


class AdaptiveHeun:
    """stepper.py:4-91: adaptive second-order scheme (Heun full step vs two half steps)."""

    def __init__(self, timeStep=1e-3, tol=1e-8, maxStep=1):
        self.dt = timeStep
        self.tolerance = tol
        self.maxStep = maxStep

    def step(self, t, f, y, normFunction=_default_norm, **rhsArgs):
        fe = 0.5
        dt = self.dt
        yInitial = y.clone()
        while fe < 1.:
            y = yInitial.clone()
            k0, info = f(y, t, **rhsArgs, intStep=0)
            y = y + dt * k0
            k1, _ = f(y, t + dt, **rhsArgs, intStep=1)
            dy0 = 0.5 * dt * (k0 + k1)
            # now with half step size (stepper.py:57-65)
            y = y - 0.5 * dt * k0
            k10, _ = f(y, t + 0.5 * dt, **rhsArgs, intStep=2)
            dy1 = 0.25 * dt * (k0 + k10)
            y = yInitial + dy1
            k01, _ = f(y, t + 0.5 * dt, **rhsArgs, intStep=3)
            y = y + 0.5 * dt * k01
            k11, _ = f(y, t + dt, **rhsArgs, intStep=4)
            dy1 = dy1 + 0.25 * dt * (k01 + k11)
            # compute deviation (stepper.py:69-72)
            updateDiff = normFunction(dy1 - dy0, f.SExp)
            fe = float(self.tolerance / updateDiff)
            if 0.2 > 0.9 * fe ** 0.33333:
                tmp = 0.2
            else:
                tmp = 0.9 * fe ** 0.33333
            if tmp > 2.:
                tmp = 2.
            realDt = dt
            dt *= tmp
            if dt > self.maxStep:
                dt = self.maxStep
        self.dt = dt
        return yInitial + dy1, realDt, info


class FixedStepper:
    """stepper.py:94-145: Heun / Euler with a geometrically growing step."""

    def __init__(self, timeStep=1e-3, maxStep=1e-2, increase_fac=1.3, mode='Heun'):
        self.dt = timeStep
        self.maxStep = maxStep
        self.mode = mode
        self.increase_fac = increase_fac

    def step(self, t, f, y, normFunction=_default_norm, **rhsArgs):
        yInitial = y.clone()
        self.dt = min(self.dt * self.increase_fac, self.maxStep)      # stepper.py:131
        if self.mode == 'Heun':
            y = yInitial.clone()
            k0, _ = f(y, t, **rhsArgs, intStep=0)
            y = y + self.dt * k0
            k1, info = f(y, t + self.dt, **rhsArgs, intStep=1)
            dy = 0.5 * self.dt * (k0 + k1)
            return yInitial + dy, self.dt, info
        if self.mode == 'Euler':
            y = yInitial.clone()
            k0, info = f(y, t, **rhsArgs, intStep=0)
            dy = self.dt * k0
            return yInitial + dy, self.dt, info
        raise ValueError(self.mode)
