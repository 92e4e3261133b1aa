"""chambolle_pock with the reference's signature (deblur_denoise/chambolle_pock/algorithm.py); see convex_optimization_amd/solvers.py."""
from ..solvers import chambolle_pock

__all__ = ["chambolle_pock"]
