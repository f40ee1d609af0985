"""ODE integrators for the parameter flow theta' = f(theta, t)  (reference: vmc_fluids/stepper.py).

The call protocol is the reference's -- `Stepper(...).step(t, f, y, normFunction, **rhsArgs) -> (y_new, dt_used, info)`,
with `f(y, t, **rhsArgs, intStep=k) -> (k_vector, info)` and, for the adaptive scheme, the attribute `f.SExp` -- but the
schemes are written here as explicit Runge-Kutta stages on new tensors: the caller's `y` is never modified (the reference
gets that for free from immutable arrays, stepper.py:136).
"""
import torch


def _default_norm(v, *_):
    return torch.linalg.norm(v)


class _Stages:
    """Evaluates the right-hand side at numbered stages and keeps the `info` of a chosen stage."""

    def __init__(self, f, rhsArgs):
        self.f, self.rhsArgs, self.count = f, rhsArgs, 0

    def __call__(self, y, t):
        k, info = self.f(y, t, **self.rhsArgs, intStep=self.count)
        self.count += 1
        return k, info


def _accept(f):
    """Before a step is handed back: let the right-hand side report a failed solve (`TDVP.check`: NaN update, Cholesky that was
    not positive definite).  The reference stops inside every call (tdvp.py:136-141); with `observables=False` the calls here do
    not synchronise with the host, so this is the one place per step that does."""
    check = getattr(f, "check", None)
    if check is not None:
        check("stepper: before accepting the step")


def _retrying(step_once):
    """A right-hand side may ask for the step to be repeated (an exception with `retry_step`: `tdvp.StaticPlanOutdated`, raised before the
    step is accepted when a plan carried over from the previous call no longer fitted -- the engine has dropped it by then).  The
    steppers do so once, from the same state and step size."""
    def step(self, t, f, y, normFunction=_default_norm, **rhsArgs):
        dt0 = self.dt
        try:
            return step_once(self, t, f, y, normFunction, **rhsArgs)
        except FloatingPointError as exc:
            if not getattr(exc, "retry_step", False):
                raise
            self.dt = dt0
            return step_once(self, t, f, y, normFunction, **rhsArgs)
    step.__doc__ = step_once.__doc__
    return step


def _heun_increment(rhs, y0, t, h, k_first=None):
    """Heun (explicit trapezoid) increment over [t, t+h] from y0; reuses a known first slope.  Returns (dy, k_first, infos)."""
    infos = []
    if k_first is None:
        k_first, i0 = rhs(y0, t)
        infos.append(i0)
    k_end, i1 = rhs(y0 + h * k_first, t + h)
    infos.append(i1)
    return 0.5 * h * (k_first + k_end), k_first, infos


class AdaptiveHeun:
    """Step-doubling error control (stepper.py:4-91): one Heun step of size dt against two of size dt/2; the difference,
    measured by `normFunction(., f.SExp)`, against `tol` decides acceptance and the next step size
    (factor 0.9 * (tol/err)^(1/3), clipped to [0.2, 2], step capped at maxStep).  Five right-hand sides per attempt, evaluated
    in the reference's order so that `intStep` numbers and `f.SExp` (the last evaluation's) agree with it."""

    SAFETY, SHRINK_MIN, GROW_MAX, EXPONENT = 0.9, 0.2, 2.0, 0.33333

    def __init__(self, timeStep=1e-3, tol=1e-8, maxStep=1):
        self.dt = timeStep
        self.tolerance = tol
        self.maxStep = maxStep

    @_retrying
    def step(self, t, f, y, normFunction=_default_norm, **rhsArgs):
        y_start = y.clone()
        dt = self.dt
        while True:
            rhs = _Stages(f, rhsArgs)
            coarse, k0, infos = _heun_increment(rhs, y_start, t, dt)                      # stages 0, 1
            info = infos[0]
            half_a, _, _ = _heun_increment(rhs, y_start, t, 0.5 * dt, k_first=k0)         # stage 2 (stage 0 reused)
            half_b, _, _ = _heun_increment(rhs, y_start + half_a, t + 0.5 * dt, 0.5 * dt)  # stages 3, 4
            fine = half_a + half_b
            # (an engine that can take the quadratic form v @ S @ v from its Jacobian in one pass hands over an operator, not the matrix)
            err = normFunction(fine - coarse, f.SExp_operator() if hasattr(f, "SExp_operator") else f.SExp)
            ratio = float(self.tolerance / err)
            used = dt
            dt = min(dt * min(max(self.SAFETY * ratio ** self.EXPONENT, self.SHRINK_MIN), self.GROW_MAX), self.maxStep)
            if ratio >= 1.0:
                break
        self.dt = dt
        _accept(f)
        return y_start + fine, used, info


class FixedStepper:
    """Heun or explicit Euler with a geometrically growing step, dt <- min(dt * increase_fac, maxStep) before every step
    (stepper.py:94-145)."""

    def __init__(self, timeStep=1e-3, maxStep=1e-2, increase_fac=1.3, mode='Heun'):
        self.dt = timeStep
        self.maxStep = maxStep
        self.mode = mode
        self.increase_fac = increase_fac

    @_retrying
    def step(self, t, f, y, normFunction=_default_norm, **rhsArgs):
        if self.mode not in ('Heun', 'Euler'):
            raise ValueError(self.mode)
        y_start = y.clone()
        self.dt = min(self.dt * self.increase_fac, self.maxStep)
        rhs = _Stages(f, rhsArgs)
        if self.mode == 'Euler':
            k0, info = rhs(y_start, t)
            _accept(f)
            return y_start + self.dt * k0, self.dt, info
        dy, _, infos = _heun_increment(rhs, y_start, t, self.dt)
        _accept(f)
        return y_start + dy, self.dt, infos[-1]
