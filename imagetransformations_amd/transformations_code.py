"""The extra transformations of the reference's later variant
(/root/reference/fall_2025/transformations_code:39-52) on the HIP kernels, same names and
argument meaning.  Its `apply_*` functions are the same bodies as `transformation.py`'s and are
re-exported from there.  `apply_perspective_warp` (:54-66) wraps torchvision's
RandomPerspective on a float tensor; torchvision is not installed, so the host side restates
its parameter draws and coefficient solve on torch's own RNG / lstsq (same generator stream,
same values) and the kernel restates `_perspective_grid` + `grid_sample`, checked bit-for-bit
against those torch CPU primitives (tests/test_oracle_vs_libs.py)."""
from __future__ import annotations

import os
import random

import numpy as np
import torch
from PIL import Image

from . import ops, staging
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


def _perspective_endpoints(width: int, height: int, distortion_scale: float):
    """RandomPerspective.get_params: eight torch.randint draws from torch's global CPU generator,
    in torchvision's order (top-left x, y; top-right; bottom-right; bottom-left)."""
    def randint(lo, hi):
        return int(torch.randint(lo, hi, size=(1,)).item())
    half_height, half_width = height // 2, width // 2
    dw, dh = int(distortion_scale * half_width), int(distortion_scale * half_height)
    topleft = [randint(0, dw + 1), randint(0, dh + 1)]
    topright = [randint(width - dw - 1, width), randint(0, dh + 1)]
    botright = [randint(width - dw - 1, width), randint(height - dh - 1, height)]
    botleft = [randint(0, dw + 1), randint(height - dh - 1, height)]
    startpoints = [[0, 0], [width - 1, 0], [width - 1, height - 1], [0, height - 1]]
    return startpoints, [topleft, topright, botright, botleft]


def _perspective_coeffs(startpoints, endpoints):
    """torchvision F._get_perspective_coeffs: fp64 least squares (gels), cast to fp32."""
    a = torch.zeros(2 * len(startpoints), 8, dtype=torch.float64)
    for i, (p1, p2) in enumerate(zip(endpoints, startpoints)):
        a[2 * i, :] = torch.tensor([p1[0], p1[1], 1, 0, 0, 0, -p2[0] * p1[0], -p2[0] * p1[1]], dtype=torch.float64)
        a[2 * i + 1, :] = torch.tensor([0, 0, 0, p1[0], p1[1], 1, -p2[1] * p1[0], -p2[1] * p1[1]], dtype=torch.float64)
    b = torch.tensor(startpoints, dtype=torch.float64).view(8)
    return torch.linalg.lstsq(a, b, driver="gels").solution.to(torch.float32).tolist()


def draw_perspective_coeffs(width: int, height: int, distortion_scale: float):
    """The random part of RandomPerspective(distortion_scale, p=1.0).forward on torch's global
    generator: the `torch.rand(1) < p` draw first (always true for p = 1), then get_params."""
    torch.rand(1)
    return _perspective_coeffs(*_perspective_endpoints(width, height, distortion_scale))


def apply_perspective_warp(img: Image.Image, distortion_scale: float = 0.2) -> Image.Image:
    """Symmetric perspective warp (:54-66): ToTensor -> RandomPerspective(distortion_scale,
    p=1.0) -> ToPILImage.  Seed with torch.manual_seed, as for the reference."""
    w, h = img.size
    coeffs = draw_perspective_coeffs(w, h, distortion_scale)
    return _download(ops.perspective(_upload(img), coeffs))


output_dir = None       # the reference hard-codes a directory (:16); None = do not save

TRANSFORMATIONS_2D = {
    'scale': {'min': 0.9, 'max': 1.4, 'step': 0.1},
    'rotation': {'min': -22.5, 'max': 22.5, 'step': 2.5},
    'lighten_darken': {'min': -0.05, 'max': 0.05, 'step': 0.01},
    'gaussian_noise': {'min': 0.0, 'max': 0.1, 'step': 0.01},
    'translation': {'min': -50, 'max': 50, 'step': 5},
    'contrast': {'min': 0, 'max': 1, 'step': 0.1},
    'blur': {'min': 0, 'max': 5, 'step': 0.5},
    'shear': {'min': 0, 'max': 1, 'step': 0.1},
    'vert_flip': {'apply': True},
    'rand_crop': {'apply': True},
    'zoom': {'min': 1.0, 'max': 1.1, 'step': 0.01},
    'perspective_warp': {'min': 0.0, 'max': 0.2, 'step': 0.05},
}

_DISPATCH = {
    'scale': apply_scale, 'rotation': apply_rotation, 'lighten_darken': apply_brightness,
    'gaussian_noise': apply_gaussian_noise, 'contrast': apply_contrast, 'shear': apply_shear,
    'blur': apply_blur, 'zoom': apply_random_zoom, 'perspective_warp': apply_perspective_warp,
}


def apply_all_transformations(images):
    """The twelve-transformation driver of the later variant (:68-155): images = [(PIL image,
    name)]; per image and per type one `random.choice` over the value grid (two for the
    translation, none for the flip and the crop), file names `{name}_{type}_{value}_corrupted.jpg`."""
    transformed_images = []
    total_transforms = 0
    for i, (img, name) in enumerate(images):
        ext = '.jpg'
        for transform_type, params in TRANSFORMATIONS_2D.items():
            if transform_type in ('vert_flip', 'rand_crop'):
                new_filename = f"{name}_{transform_type}_corrupted{ext}"
                transformed_img = vert_flip(img) if transform_type == 'vert_flip' else rand_crop(img)
            else:
                num_steps = int((params['max'] - params['min']) / params['step']) + 1
                possible_values = [params['min'] + j * params['step'] for j in range(num_steps)]
                if transform_type == 'translation':
                    tx = random.choice(possible_values)
                    ty = random.choice(possible_values)
                    new_filename = f"{name}_{transform_type}_{tx}_{ty}_corrupted{ext}"
                    transformed_img = apply_translation(img, tx, ty)
                else:
                    transform_value = random.choice(possible_values)
                    new_filename = f"{name}_{transform_type}_{transform_value}_corrupted{ext}"
                    transformed_img = _DISPATCH[transform_type](img, transform_value)
            if output_dir is not None:
                transformed_img.save(os.path.join(output_dir, new_filename))
            transformed_images.append(transformed_img)
            total_transforms += 1
        if (i + 1) % 1000 == 0:
            print(f"Processed {i + 1}/{len(images)} original images, created {total_transforms} transformed images")
    return transformed_images


def apply_all_transformations_batched(images):
    """`apply_all_transformations` with the work grouped for the GPU: the same `random`,
    `np.random` and `torch` draws in the same order, the same file names and the same pixels in
    the same output order — but every image is uploaded once and all RGB images of one size that
    drew the same (type, value) go through one batched launch (perspective warps and crops of one
    size share a launch with per-frame coefficients / windows).  images: [(PIL image, name)]."""
    from . import transformation as T
    from .transformation import (_blur_ksize, _device, _rotation_t, _scale_t, _shear_t, _translation_t)
    dev = _device()
    # ---- draws, image by image, in the order the per-image loop makes them
    plans, extra = [], {}
    for i, (img, name) in enumerate(images):
        w, h = img.size
        plan = []
        for transform_type, params in TRANSFORMATIONS_2D.items():
            if transform_type == 'vert_flip':
                plan.append((transform_type, (), f"{name}_{transform_type}_corrupted.jpg"))
            elif transform_type == 'rand_crop':
                cs = int(0.78 * w)
                extra[(i, len(plan))] = (np.random.randint(0, w - cs + 1), np.random.randint(0, h - cs + 1), cs)
                plan.append((transform_type, (), f"{name}_{transform_type}_corrupted.jpg"))
            else:
                num_steps = int((params['max'] - params['min']) / params['step']) + 1
                possible_values = [params['min'] + j * params['step'] for j in range(num_steps)]
                if transform_type == 'translation':
                    tx, ty = random.choice(possible_values), random.choice(possible_values)
                    plan.append((transform_type, (tx, ty), f"{name}_{transform_type}_{tx}_{ty}_corrupted.jpg"))
                    continue
                value = random.choice(possible_values)
                if transform_type == 'gaussian_noise':
                    shape = np.array(img).shape
                    extra[(i, len(plan))] = np.random.normal(0, value * 255, shape).astype(np.float32)
                elif transform_type == 'perspective_warp':
                    extra[(i, len(plan))] = draw_perspective_coeffs(w, h, value)
                plan.append((transform_type, (value,), f"{name}_{transform_type}_{value}_corrupted.jpg"))
        plans.append(plan)

    results = [[None] * len(p) for p in plans]
    by_size = {}
    for i, (img, _) in enumerate(images):
        if img.mode == 'RGB':
            by_size.setdefault(img.size, []).append(i)
        else:                                           # rare: the per-image bodies, with the draws made above
            for k, (transform_type, args, _) in enumerate(plans[i]):
                t = _upload(img)
                if transform_type == 'gaussian_noise':
                    results[i][k] = _download(ops.add_noise(t, torch.from_numpy(extra[(i, k)]).to(dev)))
                elif transform_type == 'perspective_warp':
                    results[i][k] = _download(ops.perspective(t, extra[(i, k)]))
                elif transform_type == 'rand_crop':
                    x, y, cs = extra[(i, k)]
                    results[i][k] = _download(ops.resize(ops.crop(t, (x, y, x + cs, y + cs)), (32, 32), ops.RESAMPLE_BICUBIC))
                elif transform_type == 'vert_flip':
                    results[i][k] = vert_flip(img)
                elif transform_type == 'translation':
                    results[i][k] = apply_translation(img, *args)
                else:
                    results[i][k] = _DISPATCH[transform_type](img, *args)

    tensor_fns = {
        'scale': _scale_t, 'zoom': _scale_t, 'rotation': _rotation_t, 'shear': _shear_t,
        'lighten_darken': lambda t, b: ops.brightness(t, 1.0 + b),
        'contrast': lambda t, a: ops.scale_abs(t, a, 0.0),
        'translation': _translation_t,
        'vert_flip': lambda t: ops.flip(t),
    }
    pending, queued = [], 0                                      # (Download, entries): results still on their way back
    for size, members in by_size.items():
        frames = staging.upload([np.asarray(images[i][0]) for i in members], dev)     # one pinned block, async H2D
        groups = {}
        for row, i in enumerate(members):
            for k, (transform_type, args, _) in enumerate(plans[i]):
                key = (transform_type, args) if transform_type not in ('perspective_warp', 'rand_crop') else (transform_type, ())
                groups.setdefault(key, []).append((row, i, k))
        for (transform_type, args), entries in groups.items():
            rows = torch.tensor([e[0] for e in entries], device=dev)
            batch = frames.index_select(0, rows)
            if transform_type == 'blur':
                ksize = _blur_ksize(args[0])
                if ksize == 0:
                    for _, i, k in entries:
                        results[i][k] = images[i][0]    # the input object itself, as the reference returns it
                    continue
                out = ops.gaussian_blur(batch, ksize, args[0], fixed_point=T.BLUR_FIXED_POINT)
            elif transform_type == 'gaussian_noise':
                z = staging.upload([extra[(i, k)] for _, i, k in entries], dev)
                out = ops.add_noise(batch, z)
            elif transform_type == 'perspective_warp':
                out = ops.perspective(batch, [extra[(i, k)] for _, i, k in entries])
            elif transform_type == 'rand_crop':
                # crops differ per image but share their size: gather them, then one resize launch
                crops = []
                for j, (_, i, k) in enumerate(entries):
                    x, y, cs = (int(v) for v in extra[(i, k)])
                    crops.append(ops.crop(batch[j], (x, y, x + cs, y + cs)))
                out = ops.resize(torch.stack(crops), (32, 32), ops.RESAMPLE_BICUBIC)
            else:
                out = tensor_fns[transform_type](batch, *args)
            pending.append((staging.download(out), entries))    # async copy back; the host waits per result below
            queued += out.numel()
            while queued > staging.PENDING_BUDGET and len(pending) > 1:      # bounded window of pinned copies in flight
                queued -= T._collect(pending.pop(0), results)

    while pending:
        T._collect(pending.pop(0), results)

    transformed_images = []
    for i, plan in enumerate(plans):
        for k, (_, _, new_filename) in enumerate(plan):
            if output_dir is not None:
                results[i][k].save(os.path.join(output_dir, new_filename))
            transformed_images.append(results[i][k])
    return transformed_images
