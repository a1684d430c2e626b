"""Batched device-tensor API over libimgxf (one HIP launch per op, no host round trips).

Every function takes uint8 ROCm tensors shaped [N,H,W,C], [H,W,C] or [H,W] (interleaved,
exactly the layout `np.array(pil_image)` has at /root/reference/transformation.py:204,229,273)
and returns a new tensor of the same rank on the same device.  Work is enqueued on the
current torch stream; nothing synchronises.  These are the batched twins of the
reference's per-image library calls — the PIL facade in `transformation.py` is a thin
H2D -> op -> D2H wrapper around them.
"""
from __future__ import annotations

import math
from typing import Sequence

import torch

from . import _ffi as F

NEAREST, BILINEAR, BICUBIC = F.FILTER_NEAREST, F.FILTER_BILINEAR, F.FILTER_BICUBIC   # affine() filter codes
# resize() filters carry Pillow's own Image.Resampling values
RESAMPLE_LANCZOS, RESAMPLE_BILINEAR, RESAMPLE_BICUBIC, RESAMPLE_BOX, RESAMPLE_HAMMING = 1, 2, 3, 4, 5
REFLECT_101, REFLECT = F.BORDER_REFLECT_101, F.BORDER_REFLECT


def _launch(t: torch.Tensor, name: str, *args) -> None:
    """Call a libimgxf entry point on `t`'s device and that device's current torch stream.

    The device comes from the tensor, not from torch's current device: a tensor on cuda:k while
    cuda:0 is current (sharding.map_frames on another rank's device, a torchrun rank that skipped
    set_device) must not have its kernel enqueued on device 0 with device-k pointers."""
    with torch.cuda.device(t.device):
        F.call(name, *args, torch.cuda.current_stream(t.device).cuda_stream)


def _check_u8(t: torch.Tensor, name: str = "image") -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(t).__name__}")
    if t.dtype != torch.uint8:
        raise TypeError(f"{name} must be uint8, got {t.dtype}")
    if not t.is_cuda:
        raise ValueError(f"{name} must live on a ROCm device (got {t.device}); libimgxf has no CPU path")
    if t.dim() not in (2, 3, 4):
        raise ValueError(f"{name} must be [N,H,W,C], [H,W,C] or [H,W]; got shape {tuple(t.shape)}")
    if t.dim() >= 3 and t.shape[-1] > 1 and t.stride(-1) != 1 or (t.dim() >= 3 and t.shape[-2] > 1 and t.stride(-2) != t.shape[-1]):
        t = t.contiguous()
    if t.dim() == 2 and t.shape[-1] > 1 and t.stride(-1) != 1:
        t = t.contiguous()
    return t


def _like(t: torch.Tensor, h: int | None = None, w: int | None = None, c: int | None = None,
          dtype=torch.uint8) -> torch.Tensor:
    """Fresh contiguous tensor with t's rank and optionally different H/W/C."""
    shape = list(t.shape)
    if t.dim() == 2:
        if h is not None: shape[0] = h
        if w is not None: shape[1] = w
        if c is not None and c != 1:
            shape = shape + [c]
    else:
        if h is not None: shape[-3] = h
        if w is not None: shape[-2] = w
        if c is not None: shape[-1] = c
    return torch.empty(shape, dtype=dtype, device=t.device)


def _out(out: torch.Tensor | None, like_shape, device, make) -> torch.Tensor:
    """The destination of an op: a fresh tensor, or the caller's pre-allocated `out` (checked)."""
    if out is None:
        return make()
    if out.dtype != torch.uint8 or out.device != device or tuple(out.shape) != tuple(like_shape):
        raise ValueError(f"out must be a uint8 tensor of shape {tuple(like_shape)} on {device}")
    return out


def _like_shape(t: torch.Tensor, h: int, w: int):
    return (h, w) if t.dim() == 2 else tuple(t.shape[:-3]) + (h, w, t.shape[-1])


def _hwc(t: torch.Tensor):
    if t.dim() == 2:
        return t.shape[0], t.shape[1], 1
    return t.shape[-3], t.shape[-2], t.shape[-1]


# ---------------------------------------------------------------- a1 Gaussian / separable
def gaussian_blur(t: torch.Tensor, ksize: int, sigma: float, return_f32: bool = False,
                  fixed_point: bool = False, out: torch.Tensor | None = None):
    """cv2.GaussianBlur(img, (ksize, ksize), sigma) — transformation.py:249.  fixed_point=True:
    OpenCV's 8-bit fixed-point evaluation (restated, unpinned) instead of the float definition.
    `out`: optional pre-allocated destination (same shape; may be a strided view)."""
    t = _check_u8(t)
    out = _out(out, t.shape, t.device, lambda: torch.empty_like(t, memory_format=torch.contiguous_format))
    if fixed_point:
        if return_f32:
            raise ValueError("the fixed-point path has no fp32 intermediate")
        _launch(t, "imgxf_gaussian_cv_fixed_u8", F.vp(F.view_of(t)), F.vp(F.view_of(out)), int(ksize), float(sigma))
        return out
    f32 = torch.empty(t.shape, dtype=torch.float32, device=t.device) if return_f32 else None
    _launch(t, "imgxf_gaussian_u8", F.vp(F.view_of(t)), F.vp(F.view_of(out)), int(ksize), float(sigma),
           F.vp(F.view_of(f32)) if return_f32 else None)
    return (out, f32) if return_f32 else out


def sepconv(t: torch.Tensor, kx: Sequence[float], ky: Sequence[float], border: int = REFLECT_101,
            return_f32: bool = False):
    t = _check_u8(t)
    out = torch.empty_like(t, memory_format=torch.contiguous_format)
    f32 = torch.empty(t.shape, dtype=torch.float32, device=t.device) if return_f32 else None
    _launch(t, "imgxf_sepconv_u8", F.vp(F.view_of(t)), F.vp(F.view_of(out)), F.f32_array(kx), len(kx),
           F.f32_array(ky), len(ky), border, F.vp(F.view_of(f32)) if return_f32 else None)
    return (out, f32) if return_f32 else out


def sepconv_fixed(t: torch.Tensor, kx: Sequence[int], ky: Sequence[int], border: int = REFLECT_101) -> torch.Tensor:
    """Separable filter with 8.8 fixed-point integer taps (each axis sums to <= 256), 16.16
    columns, (v + 2^15) >> 16 — OpenCV's uint8 evaluation order."""
    import ctypes
    t = _check_u8(t)
    out = torch.empty_like(t, memory_format=torch.contiguous_format)
    ax = (ctypes.c_uint16 * len(kx))(*[int(v) for v in kx])
    ay = (ctypes.c_uint16 * len(ky))(*[int(v) for v in ky])
    _launch(t, "imgxf_sepconv_fixed_u8", F.vp(F.view_of(t)), F.vp(F.view_of(out)), ax, len(kx), ay, len(ky), border)
    return out


# ---------------------------------------------------------------- a5 dense correlation
def conv2d(t: torch.Tensor, kernel, border: int = REFLECT_101) -> torch.Tensor:
    """cv2.filter2D(img, -1, kernel) — cifar_image_transformations.py:118."""
    t = _check_u8(t)
    rows = [list(map(float, r)) for r in kernel]
    kh, kw = len(rows), len(rows[0])
    if any(len(r) != kw for r in rows):
        raise ValueError("kernel must be rectangular")
    out = torch.empty_like(t, memory_format=torch.contiguous_format)
    flat = [v for r in rows for v in r]
    _launch(t, "imgxf_conv2d_u8", F.vp(F.view_of(t)), F.vp(F.view_of(out)), F.f32_array(flat), kh, kw,
           border)
    return out


# ---------------------------------------------------------------- a4 Sobel
def sobel(gray: torch.Tensor, variant: int = F.SOBEL_X_WRAP) -> torch.Tensor:
    """scipy.ndimage.sobel(gray_u8) — transformation.py:339 (variant 0), or |G| (variant 2)."""
    gray = _check_u8(gray, "gray")
    out = torch.empty_like(gray, memory_format=torch.contiguous_format)
    _launch(gray, "imgxf_sobel_u8", F.vp(_gray_view(gray)), F.vp(_gray_view(out)), int(variant))
    return out


def rgb_sobel_magnitude(rgb: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """Fused RGB -> L -> (Gx,Gy) -> |G| -> uint8 (benchmark configs[2])."""
    rgb = _check_u8(rgb)
    out = _out(out, tuple(rgb.shape[:-1]) + (1,) if rgb.dim() == 4 else tuple(rgb.shape[:2]), rgb.device,
               lambda: _gray_like(rgb))
    _launch(rgb, "imgxf_rgb_sobel_mag_u8", F.vp(F.view_of(rgb)), F.vp(_gray_view(out)))
    return out


def rgb_sobel(rgb: torch.Tensor, variant: int = F.SOBEL_X_WRAP) -> torch.Tensor:
    """sobel(rgb2l(rgb), variant) without materialising L (transformation.py:336-339)."""
    rgb = _check_u8(rgb)
    out = _gray_like(rgb)
    _launch(rgb, "imgxf_rgb_sobel_u8", F.vp(F.view_of(rgb)), F.vp(_gray_view(out)), int(variant))
    return out


def _gray_view(t: torch.Tensor) -> F.View:
    """View of a single-channel tensor: [H,W], [H,W,1] or [N,H,W,1]."""
    if t.dim() >= 3 and t.shape[-1] != 1:
        raise ValueError(f"expected a single-channel image, got shape {tuple(t.shape)}")
    return F.view_of(t)


def _gray_like(t: torch.Tensor) -> torch.Tensor:
    """Single-channel output for an interleaved input: [N,H,W,C] -> [N,H,W,1], [H,W,C] -> [H,W]."""
    shape = tuple(t.shape[:-1]) + (1,) if t.dim() == 4 else tuple(t.shape[:2])
    return torch.empty(shape, dtype=torch.uint8, device=t.device)


# ---------------------------------------------------------------- a2 affine family
def affine(t: torch.Tensor, matrix: Sequence[float], out_size: tuple[int, int] | None = None,
           resample: int = NEAREST, fillcolor=None, precise: bool = True, return_f32: bool = False,
           out: torch.Tensor | None = None):
    """Image.transform(out_size, AFFINE, matrix, resample, fillcolor=fillcolor).

    `out_size` is (width, height) like Pillow.  NEAREST with a pure scale/translate
    matrix follows libImaging's ImagingScaleAffine; everything else its generic path."""
    t = _check_u8(t)
    h, w, c = _hwc(t)
    ow, oh = (w, h) if out_size is None else (int(out_size[0]), int(out_size[1]))
    m = [float(v) for v in matrix][:6]
    if len(m) != 6:
        raise ValueError("affine matrix needs 6 coefficients")
    out = _out(out, _like_shape(t, oh, ow), t.device, lambda: _like(t, oh, ow))
    fill = _fill_bytes(fillcolor, c)
    if resample == NEAREST and m[1] == 0.0 and m[3] == 0.0:
        ws = torch.empty(ow + oh + 2, dtype=torch.int32, device=t.device)
        _launch(t, "imgxf_affine_scale_nearest_u8", F.vp(F.view_of(t)), F.vp(F.view_of(out)), F.f64_array(m),
               fill, ws.data_ptr(), ws.numel() * 4)
        return out
    f32 = None
    if return_f32:
        f32 = torch.empty(out.shape, dtype=torch.float32, device=t.device)
    _launch(t, "imgxf_affine_u8", F.vp(F.view_of(t)), F.vp(F.view_of(out)), F.f64_array(m), int(resample),
           fill, 1 if precise else 0, F.vp(F.view_of(f32)) if return_f32 else None)
    return (out, f32) if return_f32 else out


def _fill_bytes(fillcolor, c: int):
    if fillcolor is None:
        return F.u8_array([0] * 4)
    if isinstance(fillcolor, (int, float)):
        return F.u8_array([int(fillcolor)] * c)
    return F.u8_array(list(fillcolor)[:c])


def rotate_matrix(w: int, h: int, angle: float) -> list[float]:
    """Destination->source matrix built by Image.rotate (PIL/Image.py:2538-2568),
    expand=False, default centre, no translate."""
    cx, cy = w / 2, h / 2
    a = -math.radians(angle % 360.0)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0,
         round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    m[2] = m[0] * -cx + m[1] * -cy + m[2]
    m[5] = m[3] * -cx + m[4] * -cy + m[5]
    m[2] += cx
    m[5] += cy
    return m


def rotate_zoom_matrix(w: int, h: int, angle_deg: float, zoom: float) -> list[float]:
    """Inverse matrix for 'rotate about the centre by angle_deg and zoom' (benchmark configs[3])."""
    a = math.radians(angle_deg)
    c, s = math.cos(a) / zoom, math.sin(a) / zoom
    cx, cy = w / 2.0, h / 2.0
    return [c, -s, cx - (c * cx - s * cy), s, c, cy - (s * cx + c * cy)]


def flip(t: torch.Tensor, top_bottom: bool = False) -> torch.Tensor:
    """Image.transpose(FLIP_LEFT_RIGHT) (default) / FLIP_TOP_BOTTOM."""
    t = _check_u8(t)
    out = torch.empty_like(t, memory_format=torch.contiguous_format)
    _launch(t, "imgxf_flip_u8", F.vp(F.view_of(t)), F.vp(F.view_of(out)), 1 if top_bottom else 0)
    return out


def perspective(t: torch.Tensor, coeffs) -> torch.Tensor:
    """torchvision F.perspective(float tensor, BILINEAR, fill=0) between ToTensor and ToPILImage
    (fall_2025/transformations_code:54-66) for given coefficients: eight floats shared by all
    frames, or one row of eight per frame of a [N,H,W,C] batch."""
    t = _check_u8(t)
    rows = [list(map(float, r)) for r in coeffs] if hasattr(coeffs[0], "__len__") else None
    flat = [v for r in rows for v in r] if rows is not None else list(map(float, coeffs))
    nfr = t.shape[0] if t.dim() == 4 else 1
    if rows is not None and (t.dim() != 4 or len(rows) != nfr):
        raise ValueError("per-frame coefficients need a [N,H,W,C] batch with one row of 8 per frame")
    if len(flat) != 8 * (len(rows) if rows is not None else 1):
        raise ValueError("perspective coefficients come in rows of eight")
    out = torch.empty_like(t, memory_format=torch.contiguous_format)
    _launch(t, "imgxf_perspective_bilinear_u8", F.vp(F.view_of(t)), F.vp(F.view_of(out)), F.f32_array(flat),
           1 if rows is not None else 0)
    return out


def rot90(t: torch.Tensor, quarter_turns_ccw: int) -> torch.Tensor:
    t = _check_u8(t)
    h, w, _ = _hwc(t)
    k = quarter_turns_ccw % 4
    if k == 0:
        return t.clone()
    out = _like(t, w, h) if k != 2 else torch.empty_like(t, memory_format=torch.contiguous_format)
    _launch(t, "imgxf_rot90_u8", F.vp(F.view_of(t)), F.vp(F.view_of(out)), k)
    return out


def rotate(t: torch.Tensor, angle: float, resample: int = NEAREST, fillcolor=None,
           precise: bool = True, out: torch.Tensor | None = None) -> torch.Tensor:
    """Image.rotate(angle, resample, expand=False, fillcolor=fillcolor) incl. its fast paths."""
    t = _check_u8(t)
    h, w, _ = _hwc(t)
    a = angle % 360.0
    if a == 0 or a == 180 or (a in (90, 270) and w == h):
        res = t.clone() if a == 0 else rot90(t, 2 if a == 180 else (1 if a == 90 else 3))
        return res if out is None else out.copy_(res)
    return affine(t, rotate_matrix(w, h, angle), (w, h), resample, fillcolor, precise, out=out)


# ---------------------------------------------------------------- a3 Lanczos resize
class _PlanCache:
    """Resample plans hold only immutable device coefficient tables (created with max_frames = 0);
    the H -> V intermediate is a per-call torch tensor, i.e. stream-ordered workspace.  One cached
    plan therefore serves every stream and batch size, is never destroyed on growth (a captured HIP
    graph keeps valid pointers) and a cached plan's call neither allocates nor synchronises outside
    torch's allocator.  Plan CREATION (first use of a geometry) uploads tables synchronously and
    must happen outside stream capture."""

    def __init__(self):
        self._plans: dict = {}

    def get(self, in_h, in_w, out_h, out_w, c, device_index, resample=1, window=None):
        key = (in_h, in_w, out_h, out_w, c, device_index, resample, window)
        handle = self._plans.get(key)
        if handle is not None:
            return handle
        import ctypes
        handle = ctypes.c_void_p()
        with torch.cuda.device(device_index):
            if window is None:
                F.call("imgxf_resample_plan_create", ctypes.byref(handle), in_h, in_w, out_h, out_w, c, 0, resample)
            else:
                F.call("imgxf_resample_plan_create_window", ctypes.byref(handle), in_h, in_w, out_h, out_w, c, 0,
                       resample, *window)
        self._plans[key] = handle
        return handle

    def clear(self):
        for handle in self._plans.values():
            F.lib.imgxf_lanczos_plan_destroy(handle)
        self._plans.clear()


_plans = _PlanCache()


def _run_plan(plan, t: torch.Tensor, out: torch.Tensor, n: int) -> None:
    import ctypes
    nbytes = ctypes.c_size_t()
    vs, vo = F.view_of(t), F.view_of(out)
    F.call("imgxf_resample_workspace_bytes_for", plan, F.vp(vs), F.vp(vo), ctypes.byref(nbytes))   # 0: fused kernel
    ws = torch.empty(int(nbytes.value), dtype=torch.uint8, device=t.device) if nbytes.value else None
    with torch.cuda.device(t.device):
        F.call("imgxf_resample_ws_u8", plan, F.vp(vs), F.vp(vo), ws.data_ptr() if ws is not None else None,
               int(nbytes.value), torch.cuda.current_stream(t.device).cuda_stream)


def resize_lanczos(t: torch.Tensor, size: tuple[int, int]) -> torch.Tensor:
    """img.resize((nw, nh), Image.Resampling.LANCZOS) — transformation.py:179."""
    return resize(t, size, RESAMPLE_LANCZOS)


def resize(t: torch.Tensor, size: tuple[int, int], resample: int = RESAMPLE_BICUBIC,
           out: torch.Tensor | None = None) -> torch.Tensor:
    """Image.resize(size, resample) for the convolution filters of Resample.c (Pillow's
    Resampling values: LANCZOS=1, BILINEAR=2, BICUBIC=3 (Image.resize's default), BOX=4,
    HAMMING=5); NEAREST resize is an affine scale (`affine`).  `out`: optional destination of
    the right shape — may be a strided window of a larger tensor (e.g. a paste target)."""
    if resample not in (RESAMPLE_LANCZOS, RESAMPLE_BILINEAR, RESAMPLE_BICUBIC, RESAMPLE_BOX, RESAMPLE_HAMMING):
        raise ValueError(f"Unknown resampling filter ({resample})")
    t = _check_u8(t)
    h, w, c = _hwc(t)
    nw, nh = int(size[0]), int(size[1])
    if nw < 1 or nh < 1:
        raise ValueError("height and width must be > 0")   # Pillow's message
    n = t.shape[0] if t.dim() == 4 else 1
    if out is None:
        out = _like(t, nh, nw)
    else:
        want = [nh, nw] if t.dim() == 2 else list(t.shape[:-3]) + [nh, nw, t.shape[-1]]
        if out.dtype != torch.uint8 or out.device != t.device or tuple(out.shape) != tuple(want):
            raise ValueError(f"out must be a uint8 tensor of shape {tuple(want)} on {t.device}")
    if n == 0:
        return out
    plan = _plans.get(h, w, nh, nw, c, t.device.index or 0, int(resample))
    _run_plan(plan, t, out, n)
    return out


def resize_crop(t: torch.Tensor, size: tuple[int, int], box: tuple[int, int, int, int],
                resample: int = RESAMPLE_LANCZOS, out: torch.Tensor | None = None) -> torch.Tensor:
    """img.resize(size, resample).crop(box) without computing the cropped-away pixels: only the
    window's columns are filtered horizontally (and only the source rows its vertical taps
    touch), only its rows vertically.  Same coefficients, bit-identical to resize + crop."""
    t = _check_u8(t)
    h, w, c = _hwc(t)
    nw, nh = int(size[0]), int(size[1])
    l, tp, r, b = (int(v) for v in box)
    if not (0 <= l < r <= nw and 0 <= tp < b <= nh):
        raise ValueError("crop box must lie inside the resized image")
    if nw == w or nh == h:
        res = crop(resize(t, size, resample), box)
        return res if out is None else out.copy_(res)
    n = t.shape[0] if t.dim() == 4 else 1
    out = _out(out, _like_shape(t, b - tp, r - l), t.device, lambda: _like(t, b - tp, r - l))
    if n == 0:
        return out
    plan = _plans.get(h, w, nh, nw, c, t.device.index or 0, int(resample), (l, tp, r - l, b - tp))
    _run_plan(plan, t, out, n)
    return out


# ---------------------------------------------------------------- crop / paste / fill
def new(like: torch.Tensor, h: int, w: int, color=(0, 0, 0)) -> torch.Tensor:
    """Image.new(mode, (w,h), color) for a batch shaped like `like`."""
    _, _, c = _hwc(like)
    out = _like(like, h, w)
    _launch(out, "imgxf_fill_u8", F.vp(F.view_of(out)), _fill_bytes(color, c))
    return out


def copy_rect(src: torch.Tensor, dst: torch.Tensor, sx: int, sy: int, dx: int, dy: int, rw: int, rh: int) -> None:
    src = _check_u8(src, "src")
    _launch(src, "imgxf_copy_rect_u8", F.vp(F.view_of(src)), F.vp(F.view_of(dst)), sx, sy, dx, dy, rw, rh)


def translate(t: torch.Tensor, dx: int, dy: int, fillcolor=(0, 0, 0)) -> torch.Tensor:
    """out(x, y) = t(x - dx, y - dy) where that exists, else `fillcolor`: Image.new + crop + paste of
    apply_translation (transformation.py:284-307) in one pass."""
    t = _check_u8(t)
    _, _, c = _hwc(t)
    out = torch.empty_like(t, memory_format=torch.contiguous_format)
    if out.numel():
        _launch(t, "imgxf_translate_u8", F.vp(F.view_of(t)), F.vp(F.view_of(out)), int(dx), int(dy), _fill_bytes(fillcolor, c))
    return out


def crop(t: torch.Tensor, box: tuple[int, int, int, int]) -> torch.Tensor:
    """Image.crop((left, top, right, bottom)) for boxes inside the image."""
    l, tp, r, b = box
    out = _like(t, b - tp, r - l)
    copy_rect(t, out, l, tp, 0, 0, r - l, b - tp)
    return out


# ---------------------------------------------------------------- a6 colour maps
def rgb2l(t: torch.Tensor) -> torch.Tensor:
    """img.convert('L') — transformation.py:336."""
    t = _check_u8(t)
    if t.dim() < 3:
        raise ValueError("rgb2l expects an interleaved RGB(A) image")
    out = _gray_like(t)
    _launch(t, "imgxf_rgb2l_u8", F.vp(F.view_of(t)), F.vp(_gray_view(out)))
    return out


def scale_abs(t: torch.Tensor, alpha: float, beta: float = 0.0) -> torch.Tensor:
    """cv2.convertScaleAbs(img, alpha=alpha, beta=beta) — transformation.py:207."""
    t = _check_u8(t)
    out = torch.empty_like(t, memory_format=torch.contiguous_format)
    _launch(t, "imgxf_scale_abs_u8", F.vp(F.view_of(t)), F.vp(F.view_of(out)), float(alpha), float(beta))
    return out


def blend(im1, im2, alpha: float, like: torch.Tensor | None = None) -> torch.Tensor:
    """Image.blend(im1, im2, alpha); either image may be a solid colour tuple."""
    t1 = im1 if isinstance(im1, torch.Tensor) else None
    t2 = im2 if isinstance(im2, torch.Tensor) else None
    ref = t1 if t1 is not None else (t2 if t2 is not None else like)
    if ref is None:
        raise ValueError("blend needs at least one image tensor")
    ref = _check_u8(ref)
    _, _, c = _hwc(ref)
    if t1 is not None: t1 = _check_u8(t1, "im1")
    if t2 is not None: t2 = _check_u8(t2, "im2")
    if t1 is not None and t2 is not None and t1.shape != t2.shape:
        raise ValueError("images do not match")   # Pillow's message
    out = torch.empty_like(ref, memory_format=torch.contiguous_format)
    _launch(ref, "imgxf_blend_u8",
           F.vp(F.view_of(t1)) if t1 is not None else None, None if t1 is not None else _fill_bytes(im1, c),
           F.vp(F.view_of(t2)) if t2 is not None else None, None if t2 is not None else _fill_bytes(im2, c),
           F.vp(F.view_of(out)), float(alpha))
    return out


def brightness(t: torch.Tensor, factor: float) -> torch.Tensor:
    """ImageEnhance.Brightness(img).enhance(factor) = Image.blend(black, img, factor)."""
    return blend((0, 0, 0, 0), t, factor)


def box_blur(t: torch.Tensor, radius: float, passes: int = 1) -> torch.Tensor:
    """image.filter(ImageFilter.BoxBlur(radius)) (passes=1) — libImaging ImagingBoxBlur."""
    t = _check_u8(t)
    if radius < 0:
        raise ValueError("radius must be >= 0")        # Pillow's message
    out = torch.empty_like(t, memory_format=torch.contiguous_format)
    ws = torch.empty_like(out)
    _launch(t, "imgxf_box_blur_u8", F.vp(F.view_of(t)), F.vp(F.view_of(out)), float(radius), float(radius), int(passes),
           ws.data_ptr(), ws.numel())
    return out


def gaussian_blur_pil(t: torch.Tensor, radius: float) -> torch.Tensor:
    """image.filter(ImageFilter.GaussianBlur(radius)) — cifar_image_transformations.py:77."""
    t = _check_u8(t)
    out = torch.empty_like(t, memory_format=torch.contiguous_format)
    ws = torch.empty_like(out)
    _launch(t, "imgxf_gaussian_blur_pil_u8", F.vp(F.view_of(t)), F.vp(F.view_of(out)), float(radius),
           ws.data_ptr(), ws.numel())
    return out


SMOOTH_KERNEL = (1, 1, 1, 1, 5, 1, 1, 1, 1)      # ImageFilter.SMOOTH (scale 13)


def filter3x3(t: torch.Tensor, kernel9: Sequence[float], scale: float, offset: float = 0.0) -> torch.Tensor:
    """Image.filter(ImageFilter.Kernel((3, 3), kernel9, scale, offset)) — libImaging ImagingFilter3x3."""
    t = _check_u8(t)
    if len(kernel9) != 9:
        raise ValueError("not enough coefficients in kernel")      # Pillow's message
    out = torch.empty_like(t, memory_format=torch.contiguous_format)
    _launch(t, "imgxf_filter3x3_u8", F.vp(F.view_of(t)), F.vp(F.view_of(out)), F.f32_array(kernel9), float(scale),
           float(offset))
    return out


def enhance_sharpness(t: torch.Tensor, factor: float) -> torch.Tensor:
    """ImageEnhance.Sharpness(img).enhance(factor) = blend(img.filter(SMOOTH), img, factor)."""
    t = _check_u8(t)
    return blend(filter3x3(t, SMOOTH_KERNEL, 13.0), t, factor)


def enhance_color(t: torch.Tensor, factor: float) -> torch.Tensor:
    """ImageEnhance.Color(img).enhance(factor) — cifar_image_transformations.py:102-106."""
    t = _check_u8(t)
    out = torch.empty_like(t, memory_format=torch.contiguous_format)
    _launch(t, "imgxf_enhance_color_u8", F.vp(F.view_of(t)), F.vp(F.view_of(out)), float(factor))
    return out


def enhance_contrast(t: torch.Tensor, factor: float) -> torch.Tensor:
    """ImageEnhance.Contrast(img).enhance(factor) — cifar_image_transformations.py:81-85."""
    t = _check_u8(t)
    n = t.shape[0] if t.dim() == 4 else 1
    out = torch.empty_like(t, memory_format=torch.contiguous_format)
    sums = torch.empty(max(n, 1), dtype=torch.int64, device=t.device)
    _launch(t, "imgxf_enhance_contrast_u8", F.vp(F.view_of(t)), F.vp(F.view_of(out)), float(factor),
           sums.data_ptr())
    return out


def add_noise(t: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """np.clip(img.astype(f32) + noise, 0, 255).astype(u8) — transformation.py:275-278."""
    t = _check_u8(t)
    if noise.dtype != torch.float32 or noise.shape != t.shape or not noise.is_cuda:
        raise ValueError("noise must be a float32 device tensor shaped like the image")
    noise = noise.contiguous()
    out = torch.empty_like(t, memory_format=torch.contiguous_format)
    _launch(t, "imgxf_add_noise_u8", F.vp(F.view_of(t)), F.vp(F.view_of(noise)), F.vp(F.view_of(out)))
    return out


def add_noise_device(t: torch.Tensor, sigma: float, seed: int, offset: int = 0) -> torch.Tensor:
    """apply_gaussian_noise with the normals generated on the device (Philox4x32-10 + Box-Muller):
    clip(f32(p) + N(0, sigma), 0, 255) truncated; `sigma` = noise_std * 255.  Not NumPy's stream — the
    opt-in IMGXF_NOISE_RNG=device path of the facade.  Element e of the batch uses normal number offset + e."""
    t = _check_u8(t)
    out = torch.empty_like(t, memory_format=torch.contiguous_format)
    _launch(t, "imgxf_add_noise_philox_u8", F.vp(F.view_of(t)), F.vp(F.view_of(out)), float(sigma),
            int(seed) & 0xFFFFFFFFFFFFFFFF, int(offset))
    return out


def philox_u32(count: int, seed: int, offset: int = 0, device=None) -> torch.Tensor:
    """`count` (multiple of 4) raw Philox4x32-10 words as an int32 tensor (bit pattern of the uint32 stream)."""
    device = torch.device("cuda") if device is None else torch.device(device)
    out = torch.empty((count,), dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        F.call("imgxf_philox4x32_u32", out.data_ptr(), int(count), int(seed) & 0xFFFFFFFFFFFFFFFF, int(offset),
               torch.cuda.current_stream(device).cuda_stream)
    return out


def add_noise_f64(t: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """np.clip(img.astype(f32) + noise_f64, 0, 255).astype(u8) — cifar_image_transformations.py:45-47."""
    t = _check_u8(t)
    if noise.dtype != torch.float64 or noise.shape != t.shape or not noise.is_cuda:
        raise ValueError("noise must be a float64 device tensor shaped like the image")
    noise = noise.contiguous()
    out = torch.empty_like(t, memory_format=torch.contiguous_format)
    _launch(t, "imgxf_add_noise_f64_u8", F.vp(F.view_of(t)), F.vp(F.view_of(noise)), F.vp(F.view_of(out)))
    return out


def shot_noise_finish(counts: torch.Tensor, lam: float) -> torch.Tensor:
    """np.clip(counts / lam * 255.0, 0, 255).astype(u8) for host-drawn Poisson counts (:68-69)."""
    if counts.dtype != torch.float64 or not counts.is_cuda:
        raise ValueError("counts must be a float64 device tensor")
    counts = counts.contiguous()
    out = torch.empty(counts.shape, dtype=torch.uint8, device=counts.device)
    _launch(counts, "imgxf_shot_noise_u8", F.vp(F.view_of(counts)), float(lam), F.vp(F.view_of(out)))
    return out


def impulse_noise(t: torch.Tensor, mask: torch.Tensor, lo: float, hi: float) -> torch.Tensor:
    """img[mask < lo] = 0; img[mask > hi] = 255 for a host-drawn float64 mask [..,H,W] (:57-58)."""
    t = _check_u8(t)
    if mask.dtype != torch.float64 or not mask.is_cuda or tuple(mask.shape) != tuple(t.shape[:-1]):
        raise ValueError("mask must be a float64 device tensor shaped like the image without channels")
    mask = mask.contiguous().unsqueeze(-1)
    out = torch.empty_like(t, memory_format=torch.contiguous_format)
    _launch(t, "imgxf_impulse_noise_u8", F.vp(F.view_of(t)), F.vp(F.view_of(mask)), float(lo), float(hi),
           F.vp(F.view_of(out)))
    return out


def lut(t: torch.Tensor, table) -> torch.Tensor:
    """Image.point(table): `table` is 256 entries (all channels) or c*256 (per channel), host side."""
    t = _check_u8(t)
    c = _hwc(t)[2]
    tab = [int(v) & 0xFF for v in table]
    if len(tab) == 256:
        tab = tab * c
    if len(tab) != 256 * c:
        raise ValueError(f"lookup table needs 256 or {256 * c} entries, got {len(tab)}")
    out = torch.empty_like(t, memory_format=torch.contiguous_format)
    _launch(t, "imgxf_lut_u8", F.vp(F.view_of(t)), F.vp(F.view_of(out)), F.u8_array(tab, len(tab)))
    return out


def posterize(t: torch.Tensor, bits: int) -> torch.Tensor:
    """ImageOps.posterize (fall_2025/AugMix.py:31)."""
    mask = ~(2 ** (8 - int(bits)) - 1)
    return lut(t, [i & mask for i in range(256)])


def solarize(t: torch.Tensor, threshold: int = 128) -> torch.Tensor:
    """ImageOps.solarize (fall_2025/AugMix.py:37)."""
    return lut(t, [i if i < threshold else 255 - i for i in range(256)])


def equalize(t: torch.Tensor) -> torch.Tensor:
    """ImageOps.equalize per frame and channel (fall_2025/AugMix.py:36); histogram, table and
    mapping all run on the device."""
    t = _check_u8(t)
    h, w, c = _hwc(t)
    n = t.shape[0] if t.dim() == 4 else 1
    out = torch.empty_like(t, memory_format=torch.contiguous_format)
    ws = torch.empty(max(1, n * c * 256 * 5 // 4 + 1), dtype=torch.int32, device=t.device)
    _launch(t, "imgxf_equalize_u8", F.vp(F.view_of(t)), F.vp(F.view_of(out)), ws.data_ptr(), ws.numel() * 4)
    return out


def rgb2yuv(t: torch.Tensor) -> torch.Tensor:
    """cv2.cvtColor(img, cv2.COLOR_RGB2YUV) for 8-bit RGB (parity unpinned: OpenCV's integer definition)."""
    t = _check_u8(t)
    out = torch.empty_like(t, memory_format=torch.contiguous_format)
    _launch(t, "imgxf_rgb2yuv_u8", F.vp(F.view_of(t)), F.vp(F.view_of(out)))
    return out


def yuv2rgb(t: torch.Tensor) -> torch.Tensor:
    """cv2.cvtColor(img, cv2.COLOR_YUV2RGB) for 8-bit YUV (parity unpinned)."""
    t = _check_u8(t)
    out = torch.empty_like(t, memory_format=torch.contiguous_format)
    _launch(t, "imgxf_yuv2rgb_u8", F.vp(F.view_of(t)), F.vp(F.view_of(out)))
    return out


def equalize_hist_cv(t: torch.Tensor, channel: int = 0) -> torch.Tensor:
    """cv2.equalizeHist applied to one channel of an interleaved image, per frame (parity unpinned)."""
    t = _check_u8(t)
    h, w, c = _hwc(t)
    n = t.shape[0] if t.dim() == 4 else 1
    out = torch.empty_like(t, memory_format=torch.contiguous_format)
    ws = torch.empty(max(1, n * c * 256 * 5 // 4 + 1), dtype=torch.int32, device=t.device)
    _launch(t, "imgxf_equalize_hist_cv_u8", F.vp(F.view_of(t)), F.vp(F.view_of(out)), int(channel), ws.data_ptr(),
           ws.numel() * 4)
    return out


def channel_histogram(t: torch.Tensor) -> torch.Tensor:
    """[N, C, 256] int32 histogram of a [N,H,W,C] / [H,W,C] / [H,W] uint8 tensor."""
    t = _check_u8(t)
    h, w, c = _hwc(t)
    n = t.shape[0] if t.dim() == 4 else 1
    hist = torch.empty((n, c, 256), dtype=torch.int32, device=t.device)
    _launch(t, "imgxf_channel_histogram_u8", F.vp(F.view_of(t)), hist.data_ptr())
    return hist


def shannon_entropy(t: torch.Tensor):
    """compute_shannon_entropy (fall_2025/Initial_Experiments.py:95-113) of uint8 frames: the
    256-bin histogram over all channels comes from the device; the 256-term entropy sum is
    evaluated on the host with the reference's NumPy / SciPy expressions (density histogram,
    normalise, -sum p log p / log 2).  Returns one float per frame."""
    import numpy as np
    counts = channel_histogram(t).sum(dim=1).cpu().numpy().astype(np.float64)
    edges = np.linspace(0.0, 1.0, 257)
    out = []
    for row in counts:
        dens = row / np.diff(edges) / row.sum()
        dens = dens[dens > 0]
        pk = dens / np.sum(dens)
        out.append(float(np.sum(-pk * np.log(pk)) / np.log(2.0)))
    return out


def permute_channels(t: torch.Tensor, perm: Sequence[int]) -> torch.Tensor:
    """cv2.cvtColor channel shuffles: out[..., j] = t[..., perm[j]]."""
    t = _check_u8(t)
    out = _like(t, c=len(perm))
    _launch(t, "imgxf_permute_u8", F.vp(F.view_of(t)), F.vp(F.view_of(out)), F.i32_array(perm))
    return out


def composite(im1: torch.Tensor, im2: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """Image.composite(im1, im2, mask) for a 0/255 single-channel mask."""
    im1, im2, mask = _check_u8(im1), _check_u8(im2), _check_u8(mask, "mask")
    out = torch.empty_like(im1, memory_format=torch.contiguous_format)
    _launch(im1, "imgxf_composite_u8", F.vp(F.view_of(im1)), F.vp(F.view_of(im2)), F.vp(_gray_view(mask)),
           F.vp(F.view_of(out)))
    return out


def composite_const(im1: torch.Tensor, colour, mask: torch.Tensor) -> torch.Tensor:
    """Image.composite(im1, Image.new(mode, size, colour), mask) without the constant image."""
    im1, mask = _check_u8(im1), _check_u8(mask, "mask")
    c = _hwc(im1)[2]
    out = torch.empty_like(im1, memory_format=torch.contiguous_format)
    _launch(im1, "imgxf_composite_const_u8", F.vp(F.view_of(im1)), _fill_bytes(colour, c), F.vp(_gray_view(mask)),
           F.vp(F.view_of(out)))
    return out


# ---------------------------------------------------------------- mask stage
def percentile_mask(gray: torch.Tensor, q: float, return_threshold: bool = False):
    """(gray > np.percentile(gray, q)) * 255 per frame — transformation.py:340."""
    gray = _check_u8(gray, "gray")
    n = gray.shape[0] if gray.dim() == 4 else 1
    hist = torch.empty((n, 256), dtype=torch.int32, device=gray.device)
    thr = torch.empty(n, dtype=torch.float64, device=gray.device)
    out = torch.empty_like(gray, memory_format=torch.contiguous_format)
    gv = _gray_view(gray)
    _launch(gray, "imgxf_histogram_u8", F.vp(gv), hist.data_ptr())
    _launch(out, "imgxf_percentile_mask_u8", F.vp(gv), hist.data_ptr(), float(q), F.vp(_gray_view(out)),
           thr.data_ptr())
    return (out, thr) if return_threshold else out


def dilate_cross(mask: torch.Tensor, iterations: int) -> torch.Tensor:
    """scipy.ndimage.binary_dilation(mask, iterations=iterations) on 0/255 masks."""
    mask = _check_u8(mask, "mask")
    out = torch.empty_like(mask, memory_format=torch.contiguous_format)
    _launch(mask, "imgxf_dilate_cross_u8", F.vp(_gray_view(mask)), F.vp(_gray_view(out)), int(iterations))
    return out
