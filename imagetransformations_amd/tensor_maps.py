"""The float-tensor corruption maps of the reference's patch pipelines
(/root/reference/pipenline/angellic.py:34-46, angellic2.py:47-50) on the HIP kernel
`imgxf_f32_map`: same names, arguments and values (bit-identical to the torch expressions),
differentiable like them — the reference applies them to patched images whose patch is being
optimised, so a backward is provided (torch.clamp's rule: the gradient passes where the value
before the clamp lay in [0, 1])."""
from __future__ import annotations

import torch

from . import _ffi as F

_BRIGHTNESS, _CONTRAST, _NOISE = 0, 1, 2


def _launch(t: torch.Tensor, name: str, *args) -> None:
    """Enqueue on the tensor's own device and that device's current stream (ops._launch)."""
    with torch.cuda.device(t.device):
        F.call(name, *args, torch.cuda.current_stream(t.device).cuda_stream)


def _check(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("tensor_maps run on the HIP device only (no CPU fallback)")
    if t.dtype != torch.float32:
        raise TypeError(f"expected a float32 tensor, got {t.dtype}")
    return t.contiguous()


def _run(mode: int, x: torch.Tensor, noise, p0: float, p1: float, want_mask: bool):
    out = torch.empty_like(x)
    mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device) if want_mask else None
    _launch(x, "imgxf_f32_map", x.data_ptr(), noise.data_ptr() if noise is not None else None, out.data_ptr(),
            mask.data_ptr() if mask is not None else None, x.numel(), mode, float(p0), float(p1))
    return out, mask


class _Map(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode, noise, p0, p1):
        x = _check(x)
        noise = _check(noise) if noise is not None else None
        need = x.requires_grad
        out, mask = _run(mode, x.detach(), noise, p0, p1, need)
        if need:
            ctx.save_for_backward(mask)
            ctx.scale = p0 if mode == _CONTRAST else 1.0
        return out

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        gx = g * mask.to(g.dtype)
        if ctx.scale != 1.0:
            gx = gx * ctx.scale
        return gx, None, None, None, None


def add_gaussian_noise(images: torch.Tensor, mean: float = 0.0, std: float = 0.1) -> torch.Tensor:
    """Apply Gaussian noise to unnormalized images [0,1] (angellic.py:34-37).  The draw is
    torch.randn_like(images), as in the reference (same generator, same values)."""
    noise = torch.randn_like(images)
    return _Map.apply(images, _NOISE, noise, std, mean)


def add_brightness(images: torch.Tensor, factor: float = 0.3) -> torch.Tensor:
    """Add brightness to unnormalized images [0,1] (angellic.py:40-42)."""
    return _Map.apply(images, _BRIGHTNESS, None, factor, 0.0)


def add_contrast(images: torch.Tensor, factor: float = 1.5) -> torch.Tensor:
    """Modify contrast of unnormalized images [0,1] (angellic.py:44-46)."""
    return _Map.apply(images, _CONTRAST, None, factor, 0.0)


def to_tensor(frames: torch.Tensor, mean=None, std=None) -> torch.Tensor:
    """transforms.ToTensor() followed (when mean / std are given) by transforms.Normalize(mean, std)
    on a uint8 [N,H,W,C] / [H,W,C] / [H,W] device tensor -> float32 [N,C,H,W] / [C,H,W], in one
    pass and bit-identical to torchvision's `x.div(255)`, `sub_(mean)`, `div_(std)`."""
    if not frames.is_cuda or frames.dtype != torch.uint8:
        raise TypeError("to_tensor expects a uint8 tensor on the HIP device")
    if (mean is None) != (std is None):
        raise ValueError("mean and std come together")
    v = F.view_of(frames)
    n, h, w, c = v.n, v.h, v.w, v.c
    if mean is not None and (len(mean) != c or len(std) != c):
        raise ValueError(f"mean / std need {c} entries")
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=frames.device)
    _launch(frames, "imgxf_to_tensor_f32", F.vp(v), out.data_ptr(), F.f32_array(mean) if mean is not None else None,
            F.f32_array(std) if std is not None else None)
    return out if frames.dim() == 4 else out[0]


def resized_output_size(height: int, width: int, size: int):
    """torchvision.transforms.functional._compute_resized_output_size for an int `size` (no
    max_size): the shorter edge becomes `size`, the longer int(size * long / short)."""
    short, long = (width, height) if width <= height else (height, width)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if width <= height else (new_short, new_long)      # (new_h, new_w)


def preprocess(frames: torch.Tensor, resize: int = 256, crop: int = 224, mean=None, std=None) -> torch.Tensor:
    """transforms.Compose([Resize(resize), CenterCrop(crop), ToTensor(), Normalize(mean, std)]) on
    uint8 [N,H,W,C] / [H,W,C] device frames — the evaluation preprocessing of the reference's
    ImageNet scripts (Resize(256), CenterCrop(224)) — as Pillow / torchvision compute it on PIL
    images: BILINEAR `Image.resize` of the shorter edge (only the crop window is filtered),
    crop offsets int(round((h - crop) / 2.0)), then `to_tensor`."""
    from . import ops
    h, w = (frames.shape[-3], frames.shape[-2])
    nh, nw = resized_output_size(h, w, resize)
    if crop > nh or crop > nw:
        raise ValueError("CenterCrop larger than the resized image (torchvision pads; not needed by the reference)")
    top, left = int(round((nh - crop) / 2.0)), int(round((nw - crop) / 2.0))
    box = (left, top, left + crop, top + crop)
    if (nh, nw) == (h, w):                                   # Resize returns the image itself
        t = ops.crop(frames, box)
    else:
        t = ops.resize_crop(frames, (nw, nh), box, ops.RESAMPLE_BILINEAR)
    return to_tensor(t, mean, std)
