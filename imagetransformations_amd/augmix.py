"""AugMix operation set of the reference (/root/reference/fall_2025/AugMix.py:30-62) on the
HIP kernels: same function names, argument meaning and random-draw order.

The eight PIL -> PIL operations keep the reference's signatures `op(img, severity)`; `augmix`
takes and returns a float CHW tensor in [0, 1] like the reference, but runs the whole chain on
the device: `to_pil_image` (mul(255).byte()) and `to_tensor` (float / 255) are elementwise, so
the uint8 frame never leaves HBM between operations.  Draws come from `random` / `np.random`
in the reference's order, so a seeded run picks the same operations and weights.
"""
from __future__ import annotations

import random

import numpy as np
import torch
from PIL import Image

from . import ops
from .transformation import _download, _upload

ALPHA = 1.0  # Dirichlet / Beta parameter (AugMix.py:40)


# ---- device-tensor forms ([H,W,3] uint8) ---------------------------------------------------
def _rotate_t(t, severity):
    return ops.rotate(t, severity * random.choice([-1, 1]))              # AugMix.py:30


def _posterize_t(t, severity):
    return ops.posterize(t, int(severity))                                # :31


def _affine_t(t, data):
    h, w = t.shape[-3], t.shape[-2]
    return ops.affine(t, data, (w, h), ops.NEAREST, None)                 # Image.transform defaults


def _shear_x_t(t, severity):
    return _affine_t(t, (1, severity * 0.3, 0, 0, 1, 0))                 # :32


def _shear_y_t(t, severity):
    return _affine_t(t, (1, 0, 0, severity * 0.3, 1, 0))                 # :33


def _translate_x_t(t, severity):
    return _affine_t(t, (1, 0, severity * 2, 0, 1, 0))                   # :34


def _translate_y_t(t, severity):
    return _affine_t(t, (1, 0, 0, 0, 1, severity * 2))                   # :35


def _equalize_t(t, _):
    return ops.equalize(t)                                                # :36


def _solarize_t(t, severity):
    return ops.solarize(t, int(severity * 20))                            # :37


_TENSOR_OPS = [_rotate_t, _posterize_t, _shear_x_t, _shear_y_t, _translate_x_t, _translate_y_t,
               _equalize_t, _solarize_t]


# ---- the reference's PIL -> PIL signatures ---------------------------------------------------
def _pil(fn):
    def op(img: Image.Image, severity):
        return _download(fn(_upload(img), severity))
    op.__name__ = fn.__name__[1:-2]
    op.__doc__ = f"AugMix.py `{fn.__name__[1:-2]}(img, severity)` on the GPU."
    return op


rotate = _pil(_rotate_t)
posterize = _pil(_posterize_t)
shear_x = _pil(_shear_x_t)
shear_y = _pil(_shear_y_t)
translate_x = _pil(_translate_x_t)
translate_y = _pil(_translate_y_t)
equalize = _pil(_equalize_t)
solarize = _pil(_solarize_t)

AUG_OPS = [rotate, posterize, shear_x, shear_y, translate_x, translate_y, equalize, solarize]


_UNIT: dict = {}


def _unit_table(device) -> torch.Tensor:
    """float32 v / 255 for v = 0..255, divided on the host: torch's device kernel multiplies by
    the reciprocal, which is 1 ulp off for some v and would change the next mul(255).byte()."""
    tab = _UNIT.get(device)
    if tab is None:
        tab = _UNIT[device] = (torch.arange(256, dtype=torch.float32) / 255).to(device)
    return tab


def augmix(image_tensor: torch.Tensor, severity=3, width=3, depth=-1) -> torch.Tensor:
    """AugMix.py:45-62 for one float CHW image in [0, 1] resident on the device."""
    if not image_tensor.is_cuda:
        raise ValueError("augmix expects a device tensor (no CPU fallback)")
    ws = np.random.dirichlet([ALPHA] * width)
    m = np.random.beta(ALPHA, ALPHA)

    mix = torch.zeros_like(image_tensor)
    for i in range(width):
        image_aug = image_tensor.clone()
        d = depth if depth > 0 else np.random.randint(1, 4)
        for _ in range(d):
            k = random.choice(range(len(_TENSOR_OPS)))                    # == random.choice(AUG_OPS)
            u8 = image_aug.mul(255).byte().permute(1, 2, 0).contiguous()  # TF.to_pil_image
            u8 = _TENSOR_OPS[k](u8, severity)
            image_aug = _unit_table(u8.device)[u8.permute(2, 0, 1).long()]   # TF.to_tensor: exact v / 255
        mix += ws[i] * image_aug
    return (1 - m) * image_tensor + m * mix
