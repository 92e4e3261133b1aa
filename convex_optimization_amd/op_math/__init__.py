"""Operators and mathematical utilities (mirrors deblur_denoise/op_math/__init__.py:5-10)."""
from .operators import DeblurDenoiseOperators, apply_periodic_conv_2d, eig_vals_for_periodic_conv_op
from .apply_x_trans import apply_x_trans

__all__ = ["apply_periodic_conv_2d", "DeblurDenoiseOperators", "apply_x_trans", "eig_vals_for_periodic_conv_op"]
