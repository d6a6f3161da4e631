"""Host-side image preprocessing: ``ImageTransform`` as the callers of the reference construct it
(data/transforms.py:15-115; ``ImageTransform(1024, 512, 16)`` for the VAE path and ``ImageTransform(980, 224, 14)``
for the vision tower, app.py:71-72), written on PIL + torch (torchvision is not a dependency here).
"""
from __future__ import annotations

import numpy as np
import torch
from PIL import Image


def pil_img2rgb(image: Image.Image) -> Image.Image:
    """data/data_utils.py:118-127: flatten transparency onto white, convert to RGB."""
    if image.mode == "RGBA" or image.info.get("transparency", None) is not None:
        image = image.convert("RGBA")
        white = Image.new(mode="RGB", size=image.size, color=(255, 255, 255))
        white.paste(image, mask=image.split()[3])
        return white
    return image.convert("RGB")


class MaxLongEdgeMinShortEdgeResize:
    """data/transforms.py:15-87: bicubic resize so that the long edge is at most ``max_size``, the short edge at
    least ``min_size``, the pixel count at most ``max_pixels / img_num`` and both sides multiples of ``stride``."""

    def __init__(self, max_size: int, min_size: int, stride: int, max_pixels: int):
        self.max_size, self.min_size, self.stride, self.max_pixels = max_size, min_size, stride, max_pixels

    def _make_divisible(self, value, stride):
        return max(stride, int(round(value / stride) * stride))

    def _apply_scale(self, width, height, scale):
        return (self._make_divisible(round(width * scale), self.stride),
                self._make_divisible(round(height * scale), self.stride))

    def target_size(self, width: int, height: int, img_num: int = 1):
        scale = min(self.max_size / max(width, height), 1.0)
        scale = max(scale, self.min_size / min(width, height))
        new_width, new_height = self._apply_scale(width, height, scale)
        if new_width * new_height > self.max_pixels / img_num:
            scale = self.max_pixels / img_num / (new_width * new_height)
            new_width, new_height = self._apply_scale(new_width, new_height, scale)
        if max(new_width, new_height) > self.max_size:
            scale = self.max_size / max(new_width, new_height)
            new_width, new_height = self._apply_scale(new_width, new_height, scale)
        return new_width, new_height

    def __call__(self, img: Image.Image, img_num: int = 1) -> Image.Image:
        new_width, new_height = self.target_size(*img.size, img_num=img_num)
        return img.resize((new_width, new_height), resample=Image.BICUBIC)


class ImageTransform:
    """data/transforms.py:90-115: resize, scale to [0, 1], normalise with mean/std 0.5 -> [3, H, W] in [-1, 1]."""

    def __init__(self, max_image_size, min_image_size, image_stride, max_pixels=14 * 14 * 9 * 1024,
                 image_mean=(0.5, 0.5, 0.5), image_std=(0.5, 0.5, 0.5)):
        self.stride = image_stride
        self.resize_transform = MaxLongEdgeMinShortEdgeResize(max_image_size, min_image_size, image_stride, max_pixels)
        self.mean = torch.tensor(image_mean, dtype=torch.float32).view(-1, 1, 1)
        self.std = torch.tensor(image_std, dtype=torch.float32).view(-1, 1, 1)

    def __call__(self, img: Image.Image, img_num: int = 1) -> torch.Tensor:
        img = self.resize_transform(img, img_num=img_num)
        t = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).to(torch.float32).div(255)
        return (t - self.mean) / self.std
