"""admm_solver with the reference's signature (deblur_denoise/admm/algorithm.py); see convex_optimization_amd/solvers.py."""
from ..solvers import admm_solver

__all__ = ["admm_solver"]
