"""libImaging BoxBlur.c (ImageFilter.BoxBlur / GaussianBlur, TransformationPool.defocus_blur,
/root/reference/pipenline/cifar_image_transformations.py:72-77): the wide-lane passes against real Pillow and
against the per-byte kernel, for the radii GaussianBlur(1..5) produces, fractional box radii, every channel
count and image sizes with and without 16-byte rows."""
import numpy as np
import pytest
import torch
from PIL import Image, ImageFilter

from conftest import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hw", [(33, 64), (64, 352), (270, 480), (37, 61), (20, 16)])
def test_gaussian_and_box_blur_against_pillow(device, hw):
    from imagetransformations_amd import ops
    a = synth(500, *hw)
    t = torch.from_numpy(a).to(device)
    img = Image.fromarray(a)
    for radius in (0.5, 1, 2, 2.5, 3, 4, 5, 6, 7.3, 8, 10, 12.5):        # 3, 4, 6, 8, 10: the five defocus_blur severities
        assert np.array_equal(ops.gaussian_blur_pil(t, radius).cpu().numpy(), np.asarray(img.filter(ImageFilter.GaussianBlur(radius)))), radius
    for radius in (0.3, 1, 1.7, 2, 3.2, 4, 4.9, 6, 7.5, 9, 10.9, 11.2):
        assert np.array_equal(ops.box_blur(t, radius).cpu().numpy(), np.asarray(img.filter(ImageFilter.BoxBlur(radius)))), radius


@pytest.mark.parametrize("c", [1, 3, 4])
def test_wide_lane_passes_equal_per_byte_kernel(device, monkeypatch, c):
    from imagetransformations_amd import ops
    rng = np.random.default_rng(10 + c)
    a = rng.integers(0, 256, (3, 70, 208, c), dtype=np.uint8)
    a[0, :20] = 255; a[1, :, 50:90] = 0
    t = torch.from_numpy(a).to(device)
    for radius in (0.4, 1.0, 2.6, 3.0, 4.5, 5.2, 6.8, 8.0, 9.9, 10.5, 12.0):
        for passes in (1, 3):
            fast = ops.box_blur(t, radius, passes)
            monkeypatch.setenv("IMGXF_BOX_BYTES", "1")
            assert torch.equal(fast, ops.box_blur(t, radius, passes)), (radius, passes)
            monkeypatch.delenv("IMGXF_BOX_BYTES")
