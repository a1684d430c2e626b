"""ImagingFilter3x3 (ImageFilter.SMOOTH behind ImageEnhance.Sharpness,
/root/reference/pipenline/cifar_image_transformations.py:95-99): the 16-bytes-per-lane kernel against real
Pillow and against the per-byte kernel, for every channel count, saturating kernels and padded views."""
import numpy as np
import pytest
import torch
from PIL import Image, ImageFilter

from conftest import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hw", [(33, 64), (64, 352), (270, 480), (37, 61), (5, 16), (3, 20)])
def test_filter3x3_against_pillow(device, hw):
    from imagetransformations_amd import ops
    a = synth(400, *hw)
    t = torch.from_numpy(a).to(device)
    img = Image.fromarray(a)
    for kern, scale, off, pil in [(ops.SMOOTH_KERNEL, 13, 0, ImageFilter.SMOOTH),
                                  ((-2, -2, -2, -2, 32, -2, -2, -2, -2), 16, 0, ImageFilter.SHARPEN),
                                  ((-1, -1, -1, -1, 8, -1, -1, -1, -1), 1, 0, ImageFilter.FIND_EDGES),
                                  ((-1, 0, 0, 0, 1, 0, 0, 0, 0), 1, 128, ImageFilter.EMBOSS)]:
        got = ops.filter3x3(t, kern, scale, off).cpu().numpy()
        assert np.array_equal(got, np.asarray(img.filter(pil))), (hw, pil)


@pytest.mark.parametrize("c", [1, 3, 4])
def test_filter3x3_wide_lanes_equal_per_byte_kernel(device, monkeypatch, c):
    from imagetransformations_amd import ops
    rng = np.random.default_rng(c)
    a = rng.integers(0, 256, (3, 70, 208, c), dtype=np.uint8)
    a[0, :20] = 255; a[1, :, 50:90] = 0
    t = torch.from_numpy(a).to(device)
    big = torch.from_numpy(rng.integers(0, 256, (4, 80, 240, c), dtype=np.uint8)).to(device)
    view = big[::2, 4:76, 16:224]                                  # 16-byte aligned window of a larger batch (c = 1: 16-px offset)
    for kern, scale, off in [(ops.SMOOTH_KERNEL, 13.0, 0.0), ((0.3, -1.7, 2.2, 0.1, 0.9, -0.4, 1.5, -2.5, 0.6), 1.3, 7.5)]:
        fast = ops.filter3x3(t, kern, scale, off)
        fv = ops.filter3x3(view, kern, scale, off)
        monkeypatch.setenv("IMGXF_FILTER3X3_BYTES", "1")
        assert torch.equal(fast, ops.filter3x3(t, kern, scale, off))
        assert torch.equal(fv, ops.filter3x3(view.contiguous(), kern, scale, off))
        monkeypatch.delenv("IMGXF_FILTER3X3_BYTES")
