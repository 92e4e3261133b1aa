"""Operators and mathematical utilities (mirrors deblur_denoise/op_math/__init__.py:5-10)."""
from .operators import DeblurDenoiseOperators, apply_periodic_conv_2d, eig_vals_for_periodic_conv_op

__all__ = ["apply_periodic_conv_2d", "DeblurDenoiseOperators", "eig_vals_for_periodic_conv_op"]
