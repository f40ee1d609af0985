"""vmc_pde_amd -- MI355X-native TDVP / stochastic-reconfiguration hot path with the var_state / tdvp / stepper call
surface of RehMoritz/vmc_pde (vmc_fluids).  Compute lives in libvmcpde.so (hand-written HIP for gfx950, C ABI in
include/vmcpde.h); this package is the Python host side and has no CPU fallback."""
from . import global_defs, util, mpi_wrapper, net, sampler, var_state, evolutionEq, tdvp, stepper, grid  # noqa: F401
from .sampler import Sampler  # noqa: F401
from .var_state import VarState  # noqa: F401
from .evolutionEq import EvolutionEquation  # noqa: F401
from .tdvp import TDVP  # noqa: F401
from .stepper import FixedStepper, AdaptiveHeun  # noqa: F401

__all__ = ["Sampler", "VarState", "EvolutionEquation", "TDVP", "FixedStepper", "AdaptiveHeun"]
