"""SExp as the adaptive stepper uses it: the matrix of a norm (reference: vmc_fluids/tdvp.py:47, stepper.py:71, main.py:24-26).

`TDVP.SExp_operator()` hands the steppers a `GramOperator` in place of the P x P matrix: `v @ S @ v`, `torch.dot(v, S @ v)` and
`S @ v` are passes over the retained Jacobian; every other use gets the dense matrix `TDVP.SExp`, formed on that first use.
"""
import torch


class _VecTimesGram:
    """`v @ S` for a GramOperator S, pending: `(v @ S) @ v` -- how the reference's norm function spells the quadratic form
    (main.py:24-26) -- is then ONE pass over the Jacobian; anything else turns it into the vector S v."""

    def __init__(self, op, v):
        self._op, self._v = op, v

    def __matmul__(self, w):
        if torch.is_tensor(w) and w.dim() == 1 and (w is self._v or (w.data_ptr() == self._v.data_ptr() and w.shape == self._v.shape)):
            return self._op.quadratic_form(self._v)
        return self.vector() @ w

    def vector(self):
        return self._op.matvec(self._v)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        args = tuple(a.vector() if isinstance(a, _VecTimesGram) else a for a in args)
        return func(*args, **(kwargs or {}))

    def __getattr__(self, name):
        return getattr(self.vector(), name)


class GramOperator:
    """SExp = <(logp O~)(logp O~)^T> (tdvp.py:47) as the adaptive stepper uses it -- the matrix of a norm, `v @ S @ v` (stepper.py:71,
    main.py:24-26) -- without forming it: v^T SExp v = (1/N) sum_n (logp_n (J_n - m).v)^2 is one pass over the retained Jacobian
    (`vmcpde_jt_matvec`, first pass: 0.4 ms at the headline size) where the P x P matrix is a second Gram (17 ms).  `v @ S @ v`,
    `S @ v` and `torch.dot(v, S @ v)` take the passes; every other use (indexing, arithmetic, `.cpu()`, ...) gets the dense matrix of
    `TDVP.SExp`, formed on that first use."""

    def __init__(self, tdvp):
        self._t = tdvp

    def dense(self):
        return self._t.SExp

    def quadratic_form(self, v):
        return self._t._sexp_form(v)

    def matvec(self, v):
        return self._t._sexp_matvec(v)

    def __matmul__(self, v):
        if torch.is_tensor(v) and v.dim() == 1:
            return self.matvec(v)
        return self.dense() @ v

    def __rmatmul__(self, v):
        if torch.is_tensor(v) and v.dim() == 1:
            return _VecTimesGram(self, v)
        return v @ self.dense()

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", "")
        if name in ("matmul", "__matmul__", "mv") and len(args) == 2 and not kwargs:
            a, b = args
            if isinstance(b, GramOperator) and torch.is_tensor(a):
                return b.__rmatmul__(a)
            if isinstance(a, GramOperator) and torch.is_tensor(b):
                return a.__matmul__(b)
        if name == "__rmatmul__" and len(args) == 2 and not kwargs and isinstance(args[0], GramOperator) and torch.is_tensor(args[1]):
            return args[0].__rmatmul__(args[1])
        args = tuple(a.dense() if isinstance(a, GramOperator) else a for a in args)
        return func(*args, **kwargs)

    def __getattr__(self, name):
        return getattr(self.dense(), name)

    def __getitem__(self, idx):
        return self.dense()[idx]

