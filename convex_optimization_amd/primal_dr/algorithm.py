"""primal_dr_splitting with the reference's signature (deblur_denoise/primal_dr/algorithm.py); see convex_optimization_amd/solvers.py."""
from ..solvers import primal_dr_splitting

__all__ = ["primal_dr_splitting"]
