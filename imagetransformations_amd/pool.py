"""`TransformationPool` members that sit on the hot path, with the
reference's static-method style and argument meaning
(/root/reference/pipenline/cifar_image_transformations.py:37-129): all ten are provided.  The two
OpenCV-backed ones (`motion_blur`, `histogram_equalization`) are parity-unpinned (cv2 is not
installed where this was built; they follow OpenCV's definitions), the other eight are bit-exact
against Pillow / the reference's NumPy expressions."""
from __future__ import annotations

import random

import numpy as np
from PIL import Image

import torch

from . import ops, transformation as T
from .transformation import _device, _download, _upload


class TransformationPool:
    def histogram_equalization(image):
        """cifar_image_transformations.py:122-129: RGB -> YUV, cv2.equalizeHist on Y, YUV -> RGB.
        Parity unpinned (OpenCV is not installed where this was built): the kernels follow
        OpenCV's 8-bit integer definitions of the three calls."""
        t = _upload(image)
        return _download(ops.yuv2rgb(ops.equalize_hist_cv(ops.rgb2yuv(t), 0)))

    def gaussian_noise(image, severity=None):
        """cifar_image_transformations.py:39-48.  np.random's own stream for the same seed (computed on the device for
        images of at least NOISE_DEVICE_MIN samples, numpy_stream.py; drawn on the host below that), added and clipped on
        the device in float64."""
        if severity is None:
            severity = random.choice([1, 2, 3, 4, 5])
        img_array = np.array(image)
        noise_std = [0.08, 0.12, 0.18, 0.26, 0.38][severity - 1]
        dev = _device()
        z = None
        if T.NOISE_RNG != "numpy-host" and img_array.size >= T.NOISE_DEVICE_MIN:    # the same doubles, computed on the device
            from . import numpy_stream
            state = np.random.get_state()
            try:
                got = numpy_stream.draw_on_device([(img_array.size, noise_std * 255)], dev, f64=True)
            except ValueError:
                np.random.set_state(state)
                got = None
            z = got[0].view(img_array.shape) if got is not None else None
        if z is None:
            z = torch.from_numpy(np.random.normal(0, noise_std * 255, img_array.shape)).to(dev)
        return _download(ops.add_noise_f64(torch.from_numpy(img_array).to(dev), z))

    def impulse_noise(image, severity=None):
        """cifar_image_transformations.py:50-59."""
        if severity is None:
            severity = random.choice([1, 2, 3, 4, 5])
        img_array = np.array(image)
        noise_prob = [0.03, 0.06, 0.09, 0.17, 0.27][severity - 1]
        mask = np.random.random(img_array.shape[:2])
        dev = _device()
        out = ops.impulse_noise(torch.from_numpy(img_array).to(dev), torch.from_numpy(mask).to(dev),
                                noise_prob / 2, 1 - noise_prob / 2)
        return _download(out)

    def shot_noise(image, severity=None):
        """cifar_image_transformations.py:61-70.  The Poisson draw (whose rate is the float32
        image scaled on the host, as in the reference) stays in NumPy; scaling back, clipping
        and the uint8 cast run on the device."""
        if severity is None:
            severity = random.choice([1, 2, 3, 4, 5])
        img_array = np.array(image).astype(np.float32)
        lambda_val = [60, 25, 12, 5, 3][severity - 1]
        scaled = img_array / 255.0 * lambda_val
        counts = np.random.poisson(scaled).astype(np.float64)
        return _download(ops.shot_noise_finish(torch.from_numpy(counts).to(_device()), lambda_val))

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
