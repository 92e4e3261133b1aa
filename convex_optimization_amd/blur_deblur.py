"""End-to-end entry points with the reference's names, string dispatch and defaults
(deblur_denoise/blur_deblur.py:19-319): synthesise a blurred, noisy image, run one of the four
solvers on it, report the five end-of-run metrics.  The blur (circular_convolve2d), the solvers and
the metrics run on the device; the noise draws come from torch's CPU generator exactly as in the
reference, so a seeded run sees the same noise.  Plotting (`display`) is not provided.

Besides a path, `image_path` may be a tensor (C, H, W) or (H, W): callers that already hold the
image skip the file read."""
from __future__ import annotations

from datetime import datetime
from typing import Callable, Optional

import numpy as np
import torch

from .core.blur import create_motion_blur_kernel, gaussian_filter
from .core.convolution import circular_convolve2d
from .core.loss import l1_loss, l2_loss, mse, psnr, ssim
from .core.noise import add_gaussian_noise, add_poisson_noise, add_salt_pepper_noise, add_speckle_noise
from .solvers import admm_solver, chambolle_pock, primal_dr_splitting, primal_dual_dr_splitting
from .utils.conv_utils import read_image
from .utils.logging_utils import log_execution_time, logger

# metrics of the last blur_and_deblur_image call (bd:306-310): psnr, ssim, mse, l1_loss, l2_loss
last_metrics: dict = {}


def _load(image_path, image_shape):
    if isinstance(image_path, (torch.Tensor, np.ndarray)):
        img = torch.as_tensor(image_path)
        return img.unsqueeze(0) if img.dim() == 2 else img
    try:
        return read_image(image_path, shape=image_shape)
    except Exception as e:                                                                 # bd:45-48
        raise ValueError(f"Error reading image: {e}")


def blur_image(image_path, kernel: torch.Tensor = None, blur_type: str = "gaussian", noise_type: str = "gaussian",
               image_shape: tuple = (100, 100), blur_kernel_size: int = None, blur_kernel_sigma: float = None,
               blur_kernel_angle: float = None, salt_prob: float = None, pepper_prob: float = None, mean: float = None,
               std: float = None, scale: float = None, display: bool = False, save_path: str = None, img_id: str = None) -> torch.Tensor:
    """bd:19-81.  Returns the blurred, noisy image squeezed to (H, W)."""
    image = _load(image_path, image_shape)
    if blur_type == "gaussian":
        if kernel is None:
            kernel = gaussian_filter([blur_kernel_size, blur_kernel_size], blur_kernel_sigma)
        blurred_image = circular_convolve2d(image, kernel)
    elif blur_type == "motion":
        if kernel is None:
            kernel = create_motion_blur_kernel(blur_kernel_size, blur_kernel_angle)
        blurred_image = circular_convolve2d(image, kernel)
    elif blur_type == "none":
        blurred_image = image
    else:
        raise NotImplementedError(f"Blur type {blur_type} not implemented")

    if noise_type == "gaussian":
        noisy_image = add_gaussian_noise(blurred_image, mean, std)
    elif noise_type == "salt_pepper":
        noisy_image = add_salt_pepper_noise(blurred_image, salt_prob, pepper_prob)
    elif noise_type == "poisson":
        noisy_image = add_poisson_noise(blurred_image, scale)
    elif noise_type == "speckle":
        noisy_image = add_speckle_noise(blurred_image, std)
    elif noise_type == "none":
        noisy_image = blurred_image
    else:
        raise NotImplementedError(f"Noise type {noise_type} not implemented")
    if display:
        logger.info("display=True: plotting is not provided by this build")
    return noisy_image.squeeze()


def deblur_image(noisy_image: torch.Tensor, algorithm: str = "admm", display: bool = False, loss_function: Callable = ssim,
                 save_loss: bool = False, save_path: str = None, img_id: str = None, **kwargs) -> torch.Tensor:
    """bd:84-125: the same defaults for missing keyword arguments (t=18, rho=0.001, gamma=0.5,
    niters=1000, tol=1e-6, s=0.7) and the same argument forwarding: admm and Chambolle-Pock run
    their default l1 objective; primal_dr receives problem="primal_dr", which is not "l1" and
    therefore selects the l2 data term (pdr:101)."""
    if img_id is None:
        img_id = f"{algorithm}_{loss_function.__name__}"
    t = kwargs.get("t", 18)
    rho = kwargs.get("rho", 0.001)
    gamma = kwargs.get("gamma", 0.5)
    niters = kwargs.get("niters", 1000)
    tol = kwargs.get("tol", 1e-6)
    s = kwargs.get("s", 0.7)
    kernel = kwargs.get("kernel", None)
    extra = {k: kwargs[k] for k in ("device", "dtype", "init_state") if k in kwargs}   # this build's additions, passed through

    if algorithm == "admm":
        deblurred_image = admm_solver(b=noisy_image, kernel=kernel, t=t, rho=rho, gamma=gamma, niters=niters,
                                      loss_function=loss_function, save_loss=save_loss, **extra)
    elif algorithm == "chambolle_pock":
        deblurred_image = chambolle_pock(b=noisy_image.unsqueeze(0), kernel=kernel, t=t, s=s, gamma=gamma, max_iter=niters,
                                         loss_function=loss_function, save_loss=save_loss, **extra)
    elif algorithm == "primal_dual_dr":
        deblurred_image = primal_dual_dr_splitting(b=noisy_image, kernel=kernel, niters=niters, t=t, rho=rho, gamma=gamma, tol=tol,
                                                   loss_function=loss_function, save_loss=save_loss, **extra)
    elif algorithm == "primal_dr":
        deblurred_image = primal_dr_splitting(problem="primal_dr", b=noisy_image, kernel=kernel, t=t, gamma=gamma, rho=rho, tol=tol,
                                              niters=niters, loss_function=loss_function, save_loss=save_loss, **extra)
    else:
        raise NotImplementedError(f"Algorithm {algorithm} not implemented")
    if display:
        logger.info("display=True: plotting is not provided by this build")
    return deblurred_image


@log_execution_time(logger)
def blur_and_deblur_image(image_path, image_shape: tuple = (300, 300), experiment_id: str = None, blur_type: str = "gaussian",
                          noise_type: str = "gaussian", blur_kernel_size: int = 5, blur_kernel_sigma: float = 0.8,
                          blur_kernel_angle: float = 45, display: bool = False, kernel: torch.Tensor = None,
                          algorithm: str = "chambolle_pock", t: float = 0.4, s: float = 0.7, rho: float = 0.4, gamma: float = 0.01,
                          salt_prob: float = 0.15, pepper_prob: float = 0.15, mean: float = 0.0, std: float = 0.1, scale: float = 1.0,
                          niters: int = 1000, save_loss: bool = False, save_path: str = None, img_id: str = None,
                          loss_function: Callable = ssim) -> torch.Tensor:
    """bd:128-319.  Note (as in the reference): a caller-supplied `kernel` is used for the solve but
    not for the blur, which builds its own from blur_type and the size / sigma / angle (bd:265-277)."""
    if experiment_id is None:
        experiment_id = f'{datetime.now().strftime("%Y%m%d_%H%M%S")}'
    logger.info(f"blur_and_deblur_image experiment_id={experiment_id} blur={blur_type} noise={noise_type} algorithm={algorithm} "
                f"t={t} s={s} rho={rho} gamma={gamma} niters={niters} loss_function={getattr(loss_function, '__name__', loss_function)}")
    if img_id is None:
        img_id = f"{algorithm}_{loss_function.__name__}_{experiment_id}"
    if kernel is None:                                                                     # bd:252-261
        if blur_type == "gaussian":
            kernel = gaussian_filter(size=[blur_kernel_size, blur_kernel_size], sigma=blur_kernel_sigma)
        elif blur_type == "motion":
            kernel = create_motion_blur_kernel(size=blur_kernel_size, angle=blur_kernel_angle)
        else:
            raise ValueError(f"Unsupported blur type: {blur_type}")
        if type(kernel) == np.ndarray:
            kernel = torch.from_numpy(kernel)

    blurred_image = blur_image(image_path=image_path, image_shape=image_shape, blur_type=blur_type, noise_type=noise_type,
                               blur_kernel_size=blur_kernel_size, blur_kernel_sigma=blur_kernel_sigma,
                               blur_kernel_angle=blur_kernel_angle, salt_prob=salt_prob, pepper_prob=pepper_prob, mean=mean,
                               std=std, scale=scale, display=display)
    deblurred_image = deblur_image(blurred_image, kernel=kernel, algorithm=algorithm, t=t, s=s, rho=rho, gamma=gamma, niters=niters,
                                   display=display, loss_function=loss_function, save_loss=save_loss, save_path=save_path,
                                   img_id=img_id)
    original_image = _load(image_path, image_shape)
    if deblurred_image.shape != original_image.shape:                                      # bd:297-303
        if deblurred_image.shape[0] == 1:
            deblurred_image = deblurred_image.squeeze(0)
        elif original_image.shape[0] == 1:
            original_image = original_image.squeeze(0)
        else:
            raise ValueError(f"Deblurred image and original image have different shapes: {deblurred_image.shape} and {original_image.shape}")
    original_image = original_image.to(deblurred_image.dtype)
    # the five metrics on the device when there is one (ssim is then the fused device kernel)
    d, o = (deblurred_image.cuda(), original_image.cuda()) if torch.cuda.is_available() else (deblurred_image, original_image)
    global last_metrics
    last_metrics = {
        "psnr": float(psnr(d, o)),
        "ssim": float(torch.as_tensor(ssim(d, o)).squeeze()),
        "mse": float(mse(d, o)),
        "l1_loss": float(l1_loss(d, o)),
        "l2_loss": float(l2_loss(d, o)),
    }
    logger.info("Loss metrics w.r.t. original image: " + ", ".join(f"{k}={v}" for k, v in last_metrics.items()))
    return deblurred_image
