"""The extra transformations of the reference's later variant
(/root/reference/fall_2025/transformations_code:39-52) on the HIP kernels, same names and
argument meaning.  Its `apply_*` functions are the same bodies as `transformation.py`'s and are
re-exported from there.  `apply_perspective_warp` (:54-66) wraps torchvision's
RandomPerspective, which is not installed here, so its sampling cannot be pinned; it is not
provided (AttributeError) rather than approximated."""
from __future__ import annotations

import numpy as np
from PIL import Image

from . import ops
from .transformation import (_download, _upload, apply_blur, apply_brightness, apply_contrast,  # noqa: F401
                             apply_gaussian_noise, apply_rotation, apply_scale, apply_shear,
                             apply_translation)


def vert_flip(img: Image.Image) -> Image.Image:
    """`img.transpose(Image.FLIP_LEFT_RIGHT)` (:39-41; the reference's name notwithstanding,
    it mirrors left-right)."""
    return _download(ops.flip(_upload(img)))


def rand_crop(img: Image.Image) -> Image.Image:
    """Random crop with 0.78 scale factor, resized to 32x32 with Image.resize's default
    BICUBIC filter (:43-48).  The corner comes from np.random, as in the reference."""
    w, h = img.size
    cs = int(0.78 * w)
    x, y = np.random.randint(0, w - cs + 1), np.random.randint(0, h - cs + 1)
    t = ops.crop(_upload(img), (x, y, x + cs, y + cs))
    return _download(ops.resize(t, (32, 32), ops.RESAMPLE_BICUBIC))


def apply_random_zoom(img: Image.Image, scale_factor: float) -> Image.Image:
    """Zoom transformation (1.0 to 1.1 range) = apply_scale (:50-52)."""
    return apply_scale(img, scale_factor)
