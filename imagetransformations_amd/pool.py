"""`TransformationPool` members that sit on the hot path, with the
reference's static-method style and argument meaning
(/root/reference/pipenline/cifar_image_transformations.py:37-129).  The remaining members
(histogram equalisation, impulse / shot / float64 gaussian noise) are SURVEY §8f "next" rows and are not provided yet:
asking for them raises AttributeError rather than silently running on the CPU."""
from __future__ import annotations

import random

import numpy as np
from PIL import Image

from . import ops
from .transformation import _download, _upload


class TransformationPool:
    def motion_blur(image, size=None):
        """cifar_image_transformations.py:109-119: cv2.filter2D with a horizontal 1/size row."""
        if size is None:
            size = random.choice([5, 7, 9, 11])
        kernel = np.zeros((size, size))
        kernel[int((size - 1) / 2), :] = np.ones(size)
        kernel = kernel / size
        return _download(ops.conv2d(_upload(image), kernel.tolist()))

    def defocus_blur(image, severity=None):
        """cifar_image_transformations.py:72-77: image.filter(ImageFilter.GaussianBlur(radius))."""
        if severity is None:
            severity = random.choice([1, 2, 3, 4, 5])
        blur_levels = [3, 4, 6, 8, 10]
        radius = blur_levels[severity - 1]
        if image.mode not in ("RGB", "L"):
            raise NotImplementedError(f"defocus_blur supports RGB and L images, got {image.mode!r}")
        return _download(ops.gaussian_blur_pil(_upload(image), radius))

    def enhance_contrast(image, factor=None):
        """cifar_image_transformations.py:81-85: ImageEnhance.Contrast(image).enhance(factor)."""
        if factor is None:
            factor = random.uniform(0.5, 2.0)
        if image.mode not in ("RGB", "L"):
            raise NotImplementedError(f"enhance_contrast supports RGB and L images, got {image.mode!r}")
        return _download(ops.enhance_contrast(_upload(image), factor))

    def enhance_sharpness(image, factor=None):
        """cifar_image_transformations.py:95-99: ImageEnhance.Sharpness(image).enhance(factor)."""
        if factor is None:
            factor = random.uniform(0.5, 3.0)
        if image.mode not in ("RGB", "L"):
            raise NotImplementedError(f"enhance_sharpness supports RGB and L images, got {image.mode!r}")
        return _download(ops.enhance_sharpness(_upload(image), factor))

    def enhance_color(image, factor=None):
        """cifar_image_transformations.py:102-106: ImageEnhance.Color(image).enhance(factor)."""
        if factor is None:
            factor = random.uniform(0.5, 2.0)
        if image.mode != "RGB":
            raise NotImplementedError(f"enhance_color supports RGB images, got {image.mode!r}")
        return _download(ops.enhance_color(_upload(image), factor))

    def enhance_brightness(image, factor=None):
        """cifar_image_transformations.py:89-93: ImageEnhance.Brightness(image).enhance(factor)."""
        if factor is None:
            factor = random.uniform(0.5, 2.0)
        if image.mode not in ("RGB", "L"):
            raise NotImplementedError(f"enhance_brightness supports RGB and L images, got {image.mode!r}")
        return _download(ops.brightness(_upload(image), factor))
