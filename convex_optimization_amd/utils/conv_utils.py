"""Image input with the reference's conventions (deblur_denoise/utils/conv_utils.py:21-58): decode,
scale to [0, 1], grey-scale with the ITU-R 601 weights torchvision's rgb_to_grayscale uses, bilinear
resize with align_corners=False.  Decoding uses Pillow (the reference decodes with torchvision.io,
which is not a dependency here); the arithmetic after the decode is the reference's."""
from __future__ import annotations

import os
from typing import Optional, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F


def rgb_to_grayscale(img: torch.Tensor) -> torch.Tensor:
    """(3, H, W) -> (1, H, W) with weights 0.2989, 0.587, 0.114; a (1, H, W) image is returned as is."""
    if img.shape[0] == 1:
        return img
    if img.shape[0] < 3:
        raise ValueError(f"expected 1 or at least 3 channels, got {img.shape[0]}")
    r, g, b = img[0], img[1], img[2]
    return (0.2989 * r + 0.587 * g + 0.114 * b).to(img.dtype).unsqueeze(0)


def _pil_to_tensor_scaled(im) -> torch.Tensor:
    """What torchvision.transforms.ToTensor() returns for a PIL image (the transform the reference applies to '.tiff' files,
    cu:36-38): (C, H, W); 8-bit modes as float32 divided by 255, mode '1' as 0 / 1, modes 'I', 'I;16' and 'F' as
    int32 / int16 / float32 without scaling."""
    np_type = {"I": np.int32, "I;16": np.int16, "F": np.float32}.get(im.mode, np.uint8)
    img = torch.from_numpy(np.array(im, np_type, copy=True))
    if im.mode == "1":
        img = 255 * img
    img = img.view(im.size[1], im.size[0], len(im.getbands())).permute(2, 0, 1).contiguous()
    if img.dtype == torch.uint8:
        return img.to(torch.float32).div(255)
    return img


def read_image(path: str, device: Optional[Union[str, torch.device]] = None, shape: Optional[Tuple[int, int]] = (100, 100),
               grayscale: bool = True) -> torch.Tensor:
    """Returns a float32 (C, H, W) tensor in [0, 1] (C = 1 when grayscale)."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"Image file not found: {path}")
    from PIL import Image  # imported here: only this function needs it
    if os.path.splitext(path)[1].lower() == ".tiff":
        # cu:36-38: a '.tiff' file goes through PIL and torchvision's ToTensor(), which already scales 8-bit data to
        # [0, 1]; the division by 255 that follows (cu:43) is applied all the same, so an 8-bit TIFF ends up in [0, 1/255].
        # Reproduced as it is: it is what the reference's callers receive.
        with Image.open(path) as im:
            t = _pil_to_tensor_scaled(im)
    else:
        with Image.open(path) as im:
            if im.mode not in ("L", "RGB"):
                im = im.convert("RGB")
            arr = np.asarray(im, dtype=np.uint8)
        t = torch.from_numpy(arr.copy())
        t = t.unsqueeze(0) if t.dim() == 2 else t.permute(2, 0, 1)
    t = t.float() / 255                                                                   # cu:43
    if grayscale:
        t = rgb_to_grayscale(t)                                                           # cu:46-47
    if shape is not None:
        t = F.interpolate(t.unsqueeze(0), size=shape, mode="bilinear", align_corners=False).squeeze(0)   # cu:51-52
    if device is not None:
        t = t.to(device)
    return t
