"""CPU-only checks of the end-to-end entry points' host logic (string dispatch, defaults, image
input); the compute they drive is covered by the GPU parity tests."""
import inspect

import numpy as np
import pytest
import torch

import convex_optimization_amd as pkg
from convex_optimization_amd.utils.conv_utils import read_image, rgb_to_grayscale


def test_signatures_and_defaults_match_reference():
    """deblur_denoise/blur_deblur.py:19-34, 84-91, 129-153: names, order and defaults."""
    s = inspect.signature(pkg.blur_image)
    assert list(s.parameters) == ["image_path", "kernel", "blur_type", "noise_type", "image_shape", "blur_kernel_size", "blur_kernel_sigma",
                                  "blur_kernel_angle", "salt_prob", "pepper_prob", "mean", "std", "scale", "display", "save_path", "img_id"]
    assert s.parameters["image_shape"].default == (100, 100)
    s = inspect.signature(pkg.deblur_image)
    assert list(s.parameters) == ["noisy_image", "algorithm", "display", "loss_function", "save_loss", "save_path", "img_id", "kwargs"]
    assert s.parameters["algorithm"].default == "admm" and s.parameters["loss_function"].default is pkg.core.loss.ssim
    s = inspect.signature(inspect.unwrap(pkg.blur_and_deblur_image))
    d = {k: v.default for k, v in s.parameters.items()}
    assert (d["image_shape"], d["blur_type"], d["noise_type"], d["blur_kernel_size"], d["blur_kernel_sigma"], d["blur_kernel_angle"]) == ((300, 300), "gaussian", "gaussian", 5, 0.8, 45)
    assert (d["algorithm"], d["t"], d["s"], d["rho"], d["gamma"], d["niters"]) == ("chambolle_pock", 0.4, 0.7, 0.4, 0.01, 1000)
    assert (d["salt_prob"], d["pepper_prob"], d["mean"], d["std"], d["scale"]) == (0.15, 0.15, 0.0, 0.1, 1.0)
    assert d["loss_function"] is pkg.core.loss.ssim


def test_unknown_names_raise_like_the_reference():
    img = torch.rand(1, 8, 8)
    with pytest.raises(NotImplementedError):
        pkg.blur_image(img, blur_type="box")
    with pytest.raises(NotImplementedError):
        pkg.blur_image(img, blur_type="none", noise_type="pink")
    with pytest.raises(NotImplementedError):
        pkg.deblur_image(img[0], algorithm="fista")
    with pytest.raises(ValueError):
        pkg.blur_and_deblur_image(img, blur_type="none")       # bd:257-258: no kernel can be built
    with pytest.raises(ValueError):
        pkg.blur_image("/nonexistent/image.png")               # bd:45-48


def test_blur_image_without_blur_is_pure_host_noise():
    """blur_type 'none' touches no device code: the noise models replay torch's CPU generator."""
    img = torch.rand(1, 16, 12)
    torch.manual_seed(5)
    out = pkg.blur_image(img, blur_type="none", noise_type="gaussian", mean=0.0, std=0.1)
    torch.manual_seed(5)
    exp = torch.clamp(img + torch.randn_like(img) * 0.1 + 0.0, 0, 1).squeeze()
    assert out.shape == (16, 12) and torch.equal(out, exp)
    torch.manual_seed(5)
    out = pkg.blur_image(img, blur_type="none", noise_type="salt_pepper", salt_prob=0.1, pepper_prob=0.2)
    assert set(np.unique(out.numpy())) >= {0.0, 1.0}
    assert torch.equal(pkg.blur_image(img, blur_type="none", noise_type="none"), img.squeeze())


def test_read_image_grey_scale_and_resize(tmp_path):
    """utils/conv_utils.py:21-58: /255, weights 0.2989 / 0.587 / 0.114, bilinear align_corners=False."""
    from PIL import Image
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, size=(20, 30, 3), dtype=np.uint8)
    p = str(tmp_path / "rgb.png")
    Image.fromarray(rgb).save(p)
    t = read_image(p, shape=None)
    f = torch.from_numpy(rgb).permute(2, 0, 1).float() / 255
    exp = (0.2989 * f[0] + 0.587 * f[1] + 0.114 * f[2]).unsqueeze(0)
    assert t.shape == (1, 20, 30) and t.dtype == torch.float32 and torch.allclose(t, exp, atol=1e-7)
    t2 = read_image(p, shape=(10, 16))
    exp2 = torch.nn.functional.interpolate(exp.unsqueeze(0), size=(10, 16), mode="bilinear", align_corners=False).squeeze(0)
    assert t2.shape == (1, 10, 16) and torch.allclose(t2, exp2, atol=1e-7)
    grey = rng.integers(0, 256, size=(9, 7), dtype=np.uint8)
    pg = str(tmp_path / "grey.png")
    Image.fromarray(grey).save(pg)
    tg = read_image(pg, shape=None)
    assert tg.shape == (1, 9, 7) and torch.equal(tg, torch.from_numpy(grey).float().unsqueeze(0) / 255)
    assert read_image(p, shape=None, grayscale=False).shape == (3, 20, 30)
    assert torch.equal(rgb_to_grayscale(tg), tg)
    with pytest.raises(FileNotFoundError):
        read_image(str(tmp_path / "missing.png"))


def test_read_image_tiff_is_scaled_twice_like_the_reference(tmp_path):
    """utils/conv_utils.py:36-43: a '.tiff' file is read through PIL + ToTensor(), which scales 8-bit data to [0, 1], and
    is then divided by 255 again like every other file -- callers of the reference receive values in [0, 1/255]."""
    from PIL import Image
    rng = np.random.default_rng(1)
    grey = rng.integers(0, 256, size=(12, 9), dtype=np.uint8)
    p = str(tmp_path / "grey.tiff")
    Image.fromarray(grey).save(p)
    t = read_image(p, shape=None)
    assert t.shape == (1, 12, 9) and t.dtype == torch.float32
    assert torch.equal(t, (torch.from_numpy(grey).float().div(255) / 255).unsqueeze(0))
    rgb = rng.integers(0, 256, size=(8, 10, 3), dtype=np.uint8)
    p3 = str(tmp_path / "rgb.TIFF")
    Image.fromarray(rgb).save(p3, format="TIFF")
    f = torch.from_numpy(rgb).permute(2, 0, 1).float().div(255) / 255
    exp = (0.2989 * f[0] + 0.587 * f[1] + 0.114 * f[2]).unsqueeze(0)
    assert torch.allclose(read_image(p3, shape=None), exp, atol=1e-9)
    # the same pixels under another extension take the single scaling
    ppng = str(tmp_path / "grey.png")
    Image.fromarray(grey).save(ppng)
    assert torch.equal(read_image(ppng, shape=None), torch.from_numpy(grey).float().unsqueeze(0) / 255)

