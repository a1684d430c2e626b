"""ctypes binding of libimgxf.so (the C-ABI declared in include/imgxf.h).

There is exactly one backend: the hand-written HIP library.  If the shared object is
missing or fails to load this module raises — there is no CPU or PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("IMGXF_LIBRARY", _PKG / "libimgxf.so"))

# error codes (include/imgxf.h)
OK, ERR_NULL, ERR_SHAPE, ERR_ARG, ERR_UNSUPPORTED, ERR_WORKSPACE, ERR_NO_DEVICE = 0, -1, -2, -3, -4, -5, -6
BORDER_REFLECT_101, BORDER_REFLECT = 0, 1
FILTER_NEAREST, FILTER_BILINEAR, FILTER_BICUBIC = 0, 1, 2
SOBEL_X_WRAP, SOBEL_Y_WRAP, SOBEL_MAGNITUDE = 0, 1, 2


class ImgxfError(RuntimeError):
    """A libimgxf call failed (negative imgxf code or positive hipError_t)."""

    def __init__(self, code: int, message: str, where: str):
        super().__init__(f"{where}: {message} (code {code})")
        self.code = code


class View(C.Structure):
    """struct imgxf_view"""
    _fields_ = [("data", C.c_void_p), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("c", C.c_int32), ("row_stride", C.c_int64), ("frame_stride", C.c_int64)]


_VP = C.POINTER(View)
_F = C.POINTER(C.c_float)
_D = C.POINTER(C.c_double)
_U8 = C.POINTER(C.c_uint8)
_I32 = C.POINTER(C.c_int32)

# name -> argtypes; every function returns int except imgxf_strerror
SIGNATURES = {
    "imgxf_version": [],
    "imgxf_device_count": [],
    "imgxf_reload_knobs": [],
    "imgxf_probe_sclk": [C.c_void_p, C.c_uint, C.c_void_p],
    "imgxf_gaussian_u8": [_VP, _VP, C.c_int, C.c_double, _VP, C.c_void_p],
    "imgxf_sepconv_u8": [_VP, _VP, _F, C.c_int, _F, C.c_int, C.c_int, _VP, C.c_void_p],
    "imgxf_gaussian_cv_fixed_u8": [_VP, _VP, C.c_int, C.c_double, C.c_void_p],
    "imgxf_sepconv_fixed_u8": [_VP, _VP, C.POINTER(C.c_uint16), C.c_int, C.POINTER(C.c_uint16), C.c_int, C.c_int, C.c_void_p],
    "imgxf_conv2d_u8": [_VP, _VP, _F, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "imgxf_sobel_u8": [_VP, _VP, C.c_int, C.c_void_p],
    "imgxf_rgb_sobel_mag_u8": [_VP, _VP, C.c_void_p],
    "imgxf_rgb_sobel_u8": [_VP, _VP, C.c_int, C.c_void_p],
    "imgxf_affine_u8": [_VP, _VP, _D, C.c_int, _U8, C.c_int, _VP, C.c_void_p],
    "imgxf_affine_scale_nearest_u8": [_VP, _VP, _D, _U8, C.c_void_p, C.c_size_t, C.c_void_p],
    "imgxf_lanczos_plan_create": [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int],
    "imgxf_resample_plan_create": [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int],
    "imgxf_resample_plan_create_window": [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_int],
    "imgxf_lanczos_plan_destroy": [C.c_void_p],
    "imgxf_resize_lanczos_u8": [C.c_void_p, _VP, _VP, C.c_void_p],
    "imgxf_resample_workspace_bytes": [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)],
    "imgxf_resample_ws_u8": [C.c_void_p, _VP, _VP, C.c_void_p, C.c_size_t, C.c_void_p],
    "imgxf_resample_plan_kernel": [C.c_void_p, C.POINTER(C.c_int)],
    "imgxf_resample_workspace_bytes_for": [C.c_void_p, _VP, _VP, C.POINTER(C.c_size_t)],
    "imgxf_rgb2l_u8": [_VP, _VP, C.c_void_p],
    "imgxf_scale_abs_u8": [_VP, _VP, C.c_float, C.c_float, C.c_void_p],
    "imgxf_blend_u8": [_VP, _U8, _VP, _U8, _VP, C.c_float, C.c_void_p],
    "imgxf_add_noise_u8": [_VP, _VP, _VP, C.c_void_p],
    "imgxf_add_noise_philox_u8": [_VP, _VP, C.c_float, C.c_uint64, C.c_uint64, C.c_void_p],
    "imgxf_philox4x32_u32": [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_void_p],
    "imgxf_permute_u8": [_VP, _VP, _I32, C.c_void_p],
    "imgxf_composite_u8": [_VP, _VP, _VP, _VP, C.c_void_p],
    "imgxf_composite_const_u8": [_VP, _U8, _VP, _VP, C.c_void_p],
    "imgxf_add_noise_f64_u8": [_VP, _VP, _VP, C.c_void_p],
    "imgxf_shot_noise_u8": [_VP, C.c_double, _VP, C.c_void_p],
    "imgxf_impulse_noise_u8": [_VP, _VP, C.c_double, C.c_double, _VP, C.c_void_p],
    "imgxf_lut_u8": [_VP, _VP, _U8, C.c_void_p],
    "imgxf_equalize_u8": [_VP, _VP, C.c_void_p, C.c_size_t, C.c_void_p],
    "imgxf_channel_histogram_u8": [_VP, C.c_void_p, C.c_void_p],
    "imgxf_rgb2yuv_u8": [_VP, _VP, C.c_void_p],
    "imgxf_yuv2rgb_u8": [_VP, _VP, C.c_void_p],
    "imgxf_equalize_hist_cv_u8": [_VP, _VP, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p],
    "imgxf_box_blur_u8": [_VP, _VP, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p],
    "imgxf_gaussian_blur_pil_u8": [_VP, _VP, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p],
    "imgxf_filter3x3_u8": [_VP, _VP, _F, C.c_float, C.c_float, C.c_void_p],
    "imgxf_enhance_color_u8": [_VP, _VP, C.c_float, C.c_void_p],
    "imgxf_enhance_contrast_u8": [_VP, _VP, C.c_float, C.c_void_p, C.c_void_p],
    "imgxf_fill_u8": [_VP, _U8, C.c_void_p],
    "imgxf_copy_rect_u8": [_VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "imgxf_translate_u8": [_VP, _VP, C.c_int, C.c_int, _U8, C.c_void_p],
    "imgxf_rot90_u8": [_VP, _VP, C.c_int, C.c_void_p],
    "imgxf_flip_u8": [_VP, _VP, C.c_int, C.c_void_p],
    "imgxf_f32_map": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_void_p],
    "imgxf_to_tensor_f32": [_VP, C.c_void_p, _F, _F, C.c_void_p],
    "imgxf_perspective_bilinear_u8": [_VP, _VP, C.POINTER(C.c_float), C.c_int, C.c_void_p],
    "imgxf_histogram_u8": [_VP, C.c_void_p, C.c_void_p],
    "imgxf_jpeg_workspace_bytes": [C.c_int, C.c_int, C.c_int, C.c_size_t, C.POINTER(C.c_size_t)],
    "imgxf_jpeg_encode_u8": [_VP, C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                             C.c_void_p],
    "imgxf_np_accept": [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p],
    "imgxf_np_normals_f32": [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p],
    "imgxf_mt19937_jump": [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p],
    "imgxf_mt19937_stretches": [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p],
    "imgxf_mt19937_blocks": [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p],
    "imgxf_jpeg_unstuff_host": [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p],
    "imgxf_jpeg_layout_host": [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                               C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "imgxf_jpeg_decode_huffman": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "imgxf_jpeg_decode_idct": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    "imgxf_jpeg_decode_color": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p],
    "imgxf_percentile_mask_u8": [_VP, C.c_void_p, C.c_double, _VP, C.c_void_p, C.c_void_p],
    "imgxf_dilate_cross_u8": [_VP, _VP, C.c_int, C.c_void_p],
}


class JpegTables(C.Structure):
    """struct imgxf_jpeg_tables (include/imgxf.h)."""
    _fields_ = [("quant", (C.c_uint16 * 64) * 2), ("dc_code", (C.c_uint16 * 16) * 2), ("dc_len", (C.c_uint8 * 16) * 2),
                ("ac_code", (C.c_uint16 * 256) * 2), ("ac_len", (C.c_uint8 * 256) * 2)]


def _load() -> C.CDLL:
    # torch first: its wheel bundles the HIP runtime (libamdhip64) the process must share.  Loaded the other way round,
    # libimgxf.so binds /opt/rocm's copy, torch then brings its own, and this library's launches fail with
    # hipErrorNoDevice (seen with `python __graft_entry__.py --smoke`, where build() imports the package before torch).
    import torch  # noqa: F401
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`python -m imagetransformations_amd.build` (needs hipcc); there is no CPU fallback.")
    lib = C.CDLL(str(LIB_PATH))
    lib.imgxf_strerror.restype = C.c_char_p
    lib.imgxf_strerror.argtypes = [C.c_int]
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is not exported
        fn.restype = C.c_int
        fn.argtypes = argtypes
    return lib


lib = _load()


def strerror(code: int) -> str:
    return lib.imgxf_strerror(code).decode()


def check(code: int, where: str) -> None:
    """Map a C-ABI return code onto the exception types the reference's libraries raise:
    bad shapes/arguments -> ValueError (as Pillow/NumPy do), everything else -> ImgxfError."""
    if code == OK:
        return
    msg = strerror(code)
    if code in (ERR_SHAPE, ERR_ARG, ERR_NULL):
        raise ValueError(f"{where}: {msg} (code {code})")
    raise ImgxfError(code, msg, where)


def call(name: str, *args) -> None:
    check(getattr(lib, name)(*args), name)


def reload_knobs() -> None:
    """Re-read the IMGXF_* environment knobs (the library caches them at first use)."""
    call("imgxf_reload_knobs")


def view_of(t, elem_size: int | None = None) -> View:
    """imgxf_view of a device tensor shaped [N,H,W,C], [H,W,C] or [H,W].

    Duck-typed on `.data_ptr()`, `.shape`, `.stride()`, `.element_size()` (torch tensors);
    the innermost (channel / column) dimensions must be dense."""
    shape = tuple(t.shape)
    stride = tuple(t.stride())
    es = t.element_size() if elem_size is None else elem_size
    if len(shape) == 2:
        shape, stride = (1,) + shape + (1,), (0,) + stride + (1,)
    elif len(shape) == 3:
        shape, stride = (1,) + shape, (0,) + stride
    elif len(shape) != 4:
        raise ValueError(f"expected a [N,H,W,C], [H,W,C] or [H,W] tensor, got shape {shape}")
    n, h, w, c = shape
    if c > 1 and stride[3] != 1:
        raise ValueError("channel dimension must be contiguous (interleaved HWC layout)")
    if w > 1 and stride[2] != c:
        raise ValueError("pixels of a row must be contiguous (interleaved HWC layout)")
    row_stride = stride[1] * es if h > 1 else w * c * es
    frame_stride = stride[0] * es if n > 1 else row_stride * h
    return View(t.data_ptr(), n, h, w, c, row_stride, frame_stride)


def vp(view: View | None):
    return C.byref(view) if view is not None else None


def f32_array(values):
    arr = (C.c_float * len(values))(*[float(v) for v in values])
    return arr


def f64_array(values):
    return (C.c_double * len(values))(*[float(v) for v in values])


def u8_array(values, n=4):
    vals = [int(v) for v in values][:n]
    vals += [0] * (n - len(vals))
    return (C.c_uint8 * n)(*vals)


def i32_array(values):
    return (C.c_int32 * len(values))(*[int(v) for v in values])
