from .algorithm import admm_solver

__all__ = ["admm_solver"]
