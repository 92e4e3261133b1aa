from .algorithm import primal_dr_splitting

__all__ = ["primal_dr_splitting"]
