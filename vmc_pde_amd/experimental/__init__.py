"""Opt-in experiments around the product path of `vmc_pde_amd.tdvp` -- measured, tested against it, reported BESIDE the headline and
never as it:

    from vmc_pde_amd.experimental import TDVP
    TDVP(solver="cholesky", diagonalShift=1e-4, svdTol=1e-6, matrixFree=True)     # S never assembled (matrix_free.py)
    TDVP(solver="cholesky", diagonalShift=1e-4, svdTol=1e-6, int8Gram=True)       # S-build on the int8 matrix pipe (int8_gram.py)

With both flags off the class behaves exactly like `vmc_pde_amd.TDVP`.
"""
from dataclasses import dataclass

from .. import tdvp as _tdvp
from .int8_gram import Int8GramMixin
from .matrix_free import MatrixFreeMixin


@dataclass
class TDVP(MatrixFreeMixin, Int8GramMixin, _tdvp.TDVP):
    matrixFree: bool = False      # solver="cholesky" or the reference's spectral solve (diagonalShift = 0, no SNR), fp64 Jacobian: S is never
                                  # assembled; `S0` is then formed on first access; a full-rank S (no plan within the rank cap) raises
    int8Gram: bool = False        # the uncentred fp64 Gram (and the centred / weighted ones: SExp, rhoVar) by error-free slicing on the INT8
                                  # matrix pipe; an fp32 Jacobian takes four digits; heavy-tailed rows switch it off with a warning

    def __post_init__(self):
        self._init_matrix_free()
        self._init_int8()
        super().__post_init__()


__all__ = ["TDVP", "MatrixFreeMixin", "Int8GramMixin"]
