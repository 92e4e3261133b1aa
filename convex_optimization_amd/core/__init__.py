"""Core functionality for deblurring and denoising (mirrors deblur_denoise/core/__init__.py:5-16)."""
from .convolution import circular_convolve2d
from .blur import create_motion_blur_kernel, gaussian_filter
from .proximal_operators import prox_l1, prox_l2, prox_l2_squared, prox_box, prox_iso
from .loss import l1_loss, l2_loss, mse, psnr, ssim
from .noise import add_gaussian_noise, add_salt_pepper_noise, add_poisson_noise, add_speckle_noise

__all__ = ["circular_convolve2d", "create_motion_blur_kernel", "gaussian_filter", "prox_l1", "prox_l2",
           "prox_l2_squared", "prox_box", "prox_iso", "l1_loss", "l2_loss", "mse", "psnr", "ssim",
           "add_gaussian_noise", "add_salt_pepper_noise", "add_poisson_noise", "add_speckle_noise"]
