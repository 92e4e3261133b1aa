from .algorithm import primal_dual_dr_splitting

__all__ = ["primal_dual_dr_splitting"]
