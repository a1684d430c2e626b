"""CPU oracle for the per-pixel transform hot path (TEST INFRASTRUCTURE ONLY).

This module is a NumPy restatement of the arithmetic that the reference
(`/root/reference/transformation.py`, `/root/reference/pipenline/cifar_image_transformations.py`)
delegates to Pillow / SciPy / NumPy / OpenCV.  It is the *checker* for the HIP
kernels: only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it.  The product package `imagetransformations_amd` never
imports anything from `oracle/`.

Pinning status (see DESIGN.md "Oracle"):
  * Pillow- / SciPy- / NumPy-backed functions (rotate-nearest, affine bilinear /
    bicubic, Lanczos resize, blend / brightness, convert('L'), ndimage.sobel,
    percentile, binary_dilation, composite, translation, shear): PINNED.  They are
    checked bit-for-bit against the installed Pillow 12.2.0 / SciPy 1.15.3 called
    with the reference's own argument lists (tests/test_oracle_vs_libs.py) and
    against the committed fixtures in tests/golden/ (made by tests/golden/make_golden.py).
  * OpenCV-backed functions (cv2.GaussianBlur `transformation.py:249`,
    cv2.convertScaleAbs `transformation.py:207`, cv2.filter2D
    `cifar_image_transformations.py:118`): PARITY UNPINNED.  OpenCV (unpinned version in
    the reference's requirements) is not installed here and there is no network; the
    restatement follows OpenCV's documented definitions (Gaussian kernel formula,
    BORDER_REFLECT_101, saturate_cast round-half-even, convertScaleAbs = sat(|a*p+b|)).
    Real cv2 uses 8-bit fixed-point kernels for uint8 images, so +-1 LSB residuals
    against a real cv2 build are expected; the contract (BASELINE.json north_star) is
    the float definition to 1e-5 relative.  What does exist for them is the reference
    pipeline's own JPEG output for given parameters: tests/test_reference_outputs.py checks
    every apply_* (these three included) against those files up to JPEG noise.
  * Later additions (ImageFilter box / Gaussian blur and 3x3 kernels, ImageEnhance.*,
    ImageOps.posterize / solarize / equalize, every Resample.c filter, flips, the Pool's
    NumPy noise expressions, the entropy feature): PINNED against Pillow / NumPy / SciPy
    in tests/test_oracle_vs_libs.py.

  * torchvision-backed apply_perspective_warp (fall_2025/transformations_code:54-66):
    torchvision is not installed; its tensor path is restated here and PINNED bit for bit on
    the torch CPU primitives it consists of (linspace / bmm / grid_sample / eager arithmetic,
    tests/tv_perspective_ref.py, tests/test_oracle_vs_libs.py).  torchvision's own glue
    (get_params draw order, _get_perspective_coeffs, fill handling) is restated from its
    published source and has no installed counterpart to check against.

All image arrays are HWC (or HW) uint8, C-contiguous, RGB order — exactly what
`np.array(pil_image)` yields at `transformation.py:204,229,273`.
"""
from __future__ import annotations

import math

import numpy as np

__all__ = [name for name in dir() if not name.startswith("_")]

# ----------------------------------------------------------------------------
# borders
# ----------------------------------------------------------------------------

def reflect101_index(i, n):
    """BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba); == numpy pad mode 'reflect'.

    OpenCV default border for GaussianBlur / filter2D (transformation.py:249,
    cifar_image_transformations.py:118)."""
    i = np.asarray(i)
    if n == 1:
        return np.zeros_like(i)
    p = 2 * (n - 1)
    i = np.mod(i, p)
    return np.where(i >= n, p - i, i)


def reflect_index(i, n):
    """SciPy ndimage mode='reflect' (dcba|abcd|dcba); == numpy pad mode 'symmetric'.

    Default mode of scipy.ndimage.sobel (transformation.py:339)."""
    i = np.asarray(i)
    p = 2 * n
    i = np.mod(i, p)
    return np.where(i >= n, p - 1 - i, i)


def _as3d(img):
    img = np.asarray(img)
    if img.ndim == 2:
        return img[:, :, None], True
    return img, False


# ----------------------------------------------------------------------------
# a1: Gaussian blur  (transformation.py:228-257)  -- OpenCV-backed, parity unpinned
# ----------------------------------------------------------------------------

def blur_ksize(blur_radius):
    """Kernel-size rule of apply_blur (transformation.py:239-246).  None => no blur."""
    ksize = int(blur_radius * 6)
    if ksize % 2 == 0:
        ksize += 1
    if ksize < 3 and blur_radius > 0:
        ksize = 3
    elif blur_radius == 0:
        return None
    return ksize


# cv::getGaussianKernel's fixed kernels for sigma <= 0 and ksize in {1, 3, 5, 7} (small_gaussian_tab)
SMALL_GAUSSIAN_TAB = {
    1: (1.0,),
    3: (0.25, 0.5, 0.25),
    5: (0.0625, 0.25, 0.375, 0.25, 0.0625),
    7: (0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125),
}


def gaussian_kernel1d(ksize, sigma):
    """cv::getGaussianKernel: exp(-(i-(k-1)/2)^2 / (2 sigma^2)), normalised; sigma <= 0 takes the
    binomial table for ksize <= 7 and sigma = 0.3*((k-1)*0.5-1)+0.8 beyond (the reference always
    passes sigma = blur_radius > 0, transformation.py:249)."""
    if sigma <= 0:
        if ksize in SMALL_GAUSSIAN_TAB:
            return np.array(SMALL_GAUSSIAN_TAB[ksize], np.float64)
        sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return k / k.sum()


def sepconv_f64(img, kx, ky, index_fn=reflect101_index):
    """Separable correlation, horizontal then vertical, float64, per channel."""
    a, was2d = _as3d(img)
    h, w, _ = a.shape
    kx = np.asarray(kx, np.float64)
    ky = np.asarray(ky, np.float64)
    rx, ry = len(kx) // 2, len(ky) // 2
    src = a.astype(np.float64)
    tmp = np.zeros_like(src)
    xs = np.arange(w)
    for i, wgt in enumerate(kx):
        tmp += wgt * src[:, index_fn(xs + i - rx, w), :]
    out = np.zeros_like(src)
    ys = np.arange(h)
    for i, wgt in enumerate(ky):
        out += wgt * tmp[index_fn(ys + i - ry, h), :, :]
    return out[:, :, 0] if was2d else out


def saturate_u8(v):
    """cv::saturate_cast<uchar>(double): round half to even, clamp."""
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def gaussian_blur_f64(img, ksize, sigma):
    k = gaussian_kernel1d(ksize, sigma)
    return sepconv_f64(img, k, k)


def gaussian_blur(img, ksize, sigma):
    """cv2.GaussianBlur(img, (ksize, ksize), sigma) by its float definition."""
    return saturate_u8(gaussian_blur_f64(img, ksize, sigma))


def gaussian_kernel_cv_fixed(ksize, sigma, bits=8):
    """OpenCV's integer kernel for 8-bit images (imgproc smooth: getGaussianKernelBitExact +
    getGaussianKernelFixedPoint_ED): the float kernel times 2^bits, rounded with error diffusion
    from the ends inward, centre = 2^bits - the rest.  Restated from memory of OpenCV 4.x; double
    arithmetic stands in for its softdouble.  UNPINNED (no cv2 here) — see gaussian_blur_cv_fixed."""
    if sigma <= 0:
        if ksize in SMALL_GAUSSIAN_TAB:                       # exact multiples of 2^-bits
            return np.array([int(v * (1 << bits)) for v in SMALL_GAUSSIAN_TAB[ksize]], np.int64)
        sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    n2 = (ksize - 1) // 2
    scale2x = -0.125 / (sigma * sigma)
    vals = [math.exp((x * x) * scale2x) for x in range(1 - ksize, 0, 2)][:n2]
    mul1 = 1.0 / (2.0 * sum(vals) + 1.0)
    mult = float(1 << bits)
    err, tot, res = 0.0, 0, [0] * ksize
    for i in range(n2):
        adj = vals[i] * mul1 * mult + err
        v0 = int(np.rint(adj))
        err = adj - v0
        res[i] = res[ksize - 1 - i] = v0
        tot += v0
    res[n2] = (1 << bits) - 2 * tot
    return np.array(res, np.int64)


def gaussian_blur_cv_fixed(img, ksize, sigma):
    """What cv2.GaussianBlur most likely computes for uint8 input (OpenCV >= 4: fixed-point path):
    8.8 horizontal pass, 16.16 vertical pass, (v + 2^15) >> 16, BORDER_REFLECT_101.  Not the
    contract of the HIP kernel (BASELINE.json north_star: the float definition to 1e-5 relative)
    and not pinnable without cv2; kept because the reference's own outputs favour it: over the 69
    blur files of imagenette2/transformed it scores +0.25 dB on average and +3..4 dB on the four
    cleanest cases against the float definition, from which it differs by at most 2 LSB on 1-9 %
    of the pixels (tests/golden/validate_against_reference_outputs.py)."""
    a, was2d = _as3d(img)
    w = gaussian_kernel_cv_fixed(ksize, sigma)
    r = ksize // 2
    p = np.pad(a.astype(np.int64), ((r, r), (r, r), (0, 0)), mode="reflect")
    hp = sum(w[j] * p[:, j:j + a.shape[1]] for j in range(ksize))
    vp = sum(w[i] * hp[i:i + a.shape[0]] for i in range(ksize))
    out = np.clip((vp + 32768) >> 16, 0, 255).astype(np.uint8)
    return out[:, :, 0] if was2d else out


def apply_blur(img, blur_radius):
    """transformation.py:228-257.  The RGB<->BGR swaps (:233,:252) conjugate a
    per-channel filter and cancel; RGBA input loses alpha (:234-235)."""
    img = np.asarray(img)
    ksize = blur_ksize(blur_radius)
    if ksize is None:
        return img  # the reference returns the input object itself (:245-246)
    if img.ndim == 3 and img.shape[2] == 4:
        img = img[:, :, :3]
    return gaussian_blur(img, ksize, blur_radius)


# ----------------------------------------------------------------------------
# a5: generic 2-D correlation (cv2.filter2D), motion blur -- parity unpinned
# ----------------------------------------------------------------------------

def conv2d_f64(img, kernel, index_fn=reflect101_index):
    a, was2d = _as3d(img)
    h, w, _ = a.shape
    kernel = np.asarray(kernel, np.float64)
    kh, kw = kernel.shape
    ay, ax = kh // 2, kw // 2  # anchor (-1,-1) => centre
    src = a.astype(np.float64)
    out = np.zeros_like(src)
    ys, xs = np.arange(h), np.arange(w)
    for j in range(kh):
        rows = src[index_fn(ys + j - ay, h)]
        for i in range(kw):
            if kernel[j, i] != 0.0:
                out += kernel[j, i] * rows[:, index_fn(xs + i - ax, w), :]
    return out[:, :, 0] if was2d else out


def conv2d(img, kernel):
    """cv2.filter2D(img, -1, kernel): correlation, centre anchor, REFLECT_101."""
    return saturate_u8(conv2d_f64(img, kernel))


def motion_blur_kernel(size):
    """cifar_image_transformations.py:113-115."""
    kernel = np.zeros((size, size))
    kernel[int((size - 1) / 2), :] = np.ones(size)
    return kernel / size


def motion_blur(img, size):
    return conv2d(img, motion_blur_kernel(size))


def box_kernel(k=3):
    return np.full((k, k), 1.0 / (k * k))


# ----------------------------------------------------------------------------
# a6: elementwise colour maps
# ----------------------------------------------------------------------------

def rgb2l(img):
    """Pillow convert('L') (transformation.py:336): (19595R+38470G+7471B+0x8000)>>16."""
    a = np.asarray(img).astype(np.uint32)
    return ((a[..., 0] * 19595 + a[..., 1] * 38470 + a[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(img1, img2, alpha):
    """Pillow Image.blend (libImaging Blend.c): float32 `in1 + alpha*(in2-in1)`;
    0<=alpha<=1 truncates, otherwise clips to [0,255] and truncates."""
    a1 = np.asarray(img1).astype(np.int32)
    a2 = np.asarray(img2).astype(np.int32)
    al = np.float32(alpha)
    t = a1.astype(np.float32) + al * (a2 - a1).astype(np.float32)
    if 0.0 <= float(al) <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    out = np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int32)))
    return out.astype(np.uint8)


def apply_brightness(img, brightness_factor):
    """transformation.py:261-269: ImageEnhance.Brightness(img).enhance(1+b)
    = Image.blend(black, img, 1+b)."""
    img = np.asarray(img)
    return blend(np.zeros_like(img), img, 1.0 + brightness_factor)


def apply_background_change_simple(img, bg_color):
    """transformation.py:348-354: Image.blend(img, solid(bg), 0.3)."""
    img = np.asarray(img)
    bg = np.empty_like(img)
    bg[...] = np.array([int(c * 255) for c in bg_color], np.uint8)
    return blend(img, bg, 0.3)


def convert_scale_abs(img, alpha, beta=0.0):
    """cv2.convertScaleAbs: saturate_cast<uchar>(|alpha*p + beta|) (float32 math)."""
    a = np.asarray(img).astype(np.float32)
    v = np.abs(a * np.float32(alpha) + np.float32(beta))
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def apply_contrast(img, contrast_amount):
    """transformation.py:203-210 (RGBA loses alpha at :205-206)."""
    img = np.asarray(img)
    if img.ndim == 3 and img.shape[2] == 4:
        img = img[:, :, :3]
    return convert_scale_abs(img, contrast_amount, 0.0)


def add_noise(img, noise_f32):
    """transformation.py:275-278 with the noise tensor given:
    clip(f32(p)+noise, 0, 255).astype(u8) (truncation)."""
    a = np.asarray(img).astype(np.float32) + np.asarray(noise_f32, np.float32)
    return np.clip(a, 0, 255).astype(np.uint8)


def apply_gaussian_noise(img, noise_std, rng=None):
    """transformation.py:272-281.  `rng` None => NumPy global MT19937 as the reference."""
    img = np.asarray(img)
    gen = np.random if rng is None else rng
    noise = gen.normal(0, noise_std * 255, img.shape).astype(np.float32)
    return add_noise(img, noise)


def permute_channels(img, perm):
    """cv2.cvtColor RGB2BGR/BGR2RGB/RGBA2RGB/RGBA2BGR (transformation.py:206,233-235,252)."""
    return np.ascontiguousarray(np.asarray(img)[..., list(perm)])


def filter3x3(img, kernel9, scale, offset=0.0):
    """Image.filter(ImageFilter.Kernel((3,3), kernel9, scale, offset)) — libImaging
    ImagingFilter3x3: float32 coefficients kernel/scale, ss = offset + 0.5, then one
    `(a*k0 + b*k1) + c*k2` per row (row y+1 first), clip8 truncation; the 1-pixel frame
    is copied from the input."""
    a = np.asarray(img)
    k = (np.asarray(kernel9, np.float32) / np.float32(scale)).astype(np.float32)
    h, w = a.shape[:2]
    out = a.copy()
    if h < 3 or w < 3:
        return out
    f = a.astype(np.float32)

    def row3(r, kk):
        return (r[:, :-2] * kk[0] + r[:, 1:-1] * kk[1]) + r[:, 2:] * kk[2]

    ss = (np.float32(offset) + np.float32(0.5)) + row3(f[2:], k[0:3])
    ss = ss + row3(f[1:-1], k[3:6])
    ss = ss + row3(f[:-2], k[6:9])
    q = np.where(ss <= 0, 0, np.where(ss >= 255, 255, ss.astype(np.int32))).astype(np.uint8)
    out[1:-1, 1:-1] = q
    return out


SMOOTH_KERNEL = (1, 1, 1, 1, 5, 1, 1, 1, 1)   # ImageFilter.SMOOTH, scale 13


def gaussian_box_radius(radius, passes=3):
    """libImaging BoxBlur.c _gaussian_blur_radius: float32 variables, double sqrt/floor."""
    f32 = np.float32
    radius = f32(radius)
    sigma2 = f32(radius * radius / f32(passes))
    L = f32(math.sqrt(12.0 * float(sigma2) + 1.0))
    l = f32(math.floor((float(L) - 1.0) / 2.0))
    a = f32(f32(f32(2) * l + f32(1)) * f32(f32(l * f32(l + f32(1))) - f32(f32(3) * sigma2)))
    a = f32(a / f32(f32(6) * f32(sigma2 - f32(f32(l + f32(1)) * f32(l + f32(1))))))
    return f32(l + a)


def box_blur_pass(img, float_radius, axis):
    """One ImagingHorizontalBoxBlur pass along `axis` (1 = x, 0 = y): exact uint32 arithmetic,
    edge pixels replicated, out = (window_sum*ww + (far_left+far_right)*fw + 2^23) >> 24."""
    a = np.asarray(img)
    fr = np.float32(float_radius)
    radius = int(fr)
    ww = int(np.float32(16777216.0) / np.float32(fr * np.float32(2) + np.float32(1))) & 0xFFFFFFFF
    fw = ((16777216 - (radius * 2 + 1) * ww) & 0xFFFFFFFF) // 2
    n = a.shape[axis]
    idx = np.arange(n)
    src = np.moveaxis(a, axis, 0).astype(np.uint64)
    acc = np.zeros_like(src)
    for i in range(-radius, radius + 1):
        acc += src[np.clip(idx + i, 0, n - 1)]
    far = src[np.clip(idx - radius - 1, 0, n - 1)] + src[np.clip(idx + radius + 1, 0, n - 1)]
    bulk = (acc * ww + far * fw) & 0xFFFFFFFF
    out = (((bulk + (1 << 23)) & 0xFFFFFFFF) >> 24).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def box_blur(img, xradius, yradius, passes=1):
    """ImagingBoxBlur: `passes` horizontal passes, then `passes` vertical ones (ImageFilter.BoxBlur)."""
    out = np.asarray(img)
    if xradius != 0:
        for _ in range(passes):
            out = box_blur_pass(out, xradius, 1)
    if yradius != 0:
        for _ in range(passes):
            out = box_blur_pass(out, yradius, 0)
    return out if out is not img else np.array(img)


def pil_gaussian_blur(img, radius):
    """image.filter(ImageFilter.GaussianBlur(radius)) — TransformationPool.defocus_blur
    (cifar_image_transformations.py:72-77): three box-blur passes per axis."""
    r = gaussian_box_radius(radius, 3)
    return box_blur(img, r, r, 3)


def enhance_sharpness(img, factor):
    """ImageEnhance.Sharpness(img).enhance(factor) (cifar_image_transformations.py:95-99):
    Image.blend(img.filter(ImageFilter.SMOOTH), img, factor)."""
    a = np.asarray(img)
    return blend(filter3x3(a, SMOOTH_KERNEL, 13), a, factor)


def enhance_color(img, factor):
    """ImageEnhance.Color(img).enhance(factor) (cifar_image_transformations.py:102-106):
    degenerate = img.convert('L').convert('RGB'); Image.blend(degenerate, img, factor)."""
    a = np.asarray(img)
    g = rgb2l(a)
    return blend(np.repeat(g[:, :, None], 3, axis=2), a, factor)


def enhance_contrast(img, factor):
    """ImageEnhance.Contrast(img).enhance(factor) (cifar_image_transformations.py:81-85):
    degenerate = solid int(mean(L) + 0.5); Image.blend(degenerate, img, factor)."""
    a = np.asarray(img)
    g = rgb2l(a) if a.ndim == 3 else a
    mean = int(int(g.astype(np.int64).sum()) / g.size + 0.5)
    deg = np.full_like(a, mean)
    return blend(deg, a, factor)


# ----------------------------------------------------------------------------
# a2 / a2' / shear: Pillow affine transform (libImaging Geometry.c)
# ----------------------------------------------------------------------------

def rotate_plan(w, h, angle):
    """Python layer of Image.rotate(angle) (PIL/Image.py:2509-2568), expand=False,
    default centre.  Returns ("copy"|"rot180"|"rot90"|"rot270", None) for the fast
    paths or ("affine", m[6]) with the destination->source matrix."""
    angle = angle % 360.0
    if angle == 0:
        return "copy", None
    if angle == 180:
        return "rot180", None
    if angle in (90, 270) and w == h:
        return ("rot90" if angle == 90 else "rot270"), None
    cx, cy = w / 2, h / 2
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0,
         round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    m[2] = m[0] * -cx + m[1] * -cy + m[2]
    m[5] = m[3] * -cx + m[4] * -cy + m[5]
    m[2] += cx
    m[5] += cy
    return "affine", m


def rotate_zoom_matrix(w, h, angle_deg, zoom):
    """Inverse matrix for 'rotate by angle about the centre and zoom by `zoom`'
    (benchmark configs[3]; SURVEY §8a row a2')."""
    a = math.radians(angle_deg)
    c, s = math.cos(a) / zoom, math.sin(a) / zoom
    cx, cy = w / 2.0, h / 2.0
    return [c, -s, cx - (c * cx - s * cy), s, c, cy - (s * cx + c * cy)]


def _fix16(v):
    return int(math.floor(v * 65536.0 + 0.5))


def _fill_array(oh, ow, c, fill, dtype=np.uint8):
    out = np.empty((oh, ow, c), dtype)
    f = np.zeros(c, dtype) if fill is None else np.asarray(list(fill)[:c], dtype)
    out[...] = f
    return out


def affine_nearest(img, out_size, m, fill=None):
    """Image.transform(size, AFFINE, m, NEAREST, fillcolor=fill).

    Rotations/shears (m1 or m3 non-zero) go through `affine_fixed` (16.16 fixed
    point); pure scale/translate matrices go through `ImagingScaleAffine`, which
    walks the source coordinate by repeated double additions."""
    a, was2d = _as3d(img)
    h, w, c = a.shape
    ow, oh = out_size
    out = _fill_array(oh, ow, c, fill)
    if m[1] == 0 and m[3] == 0:
        # ImagingScaleAffine: xo = a2 + a0*0.5; xo += a0 per pixel; COORD(v) = v<0 ? -1 : (int)v
        xo = m[2] + m[0] * 0.5
        xin = np.empty(ow, np.int64)
        for x in range(ow):
            xin[x] = -1 if xo < 0.0 else int(xo)
            xo += m[0]
        yo = m[5] + m[4] * 0.5
        yin = np.empty(oh, np.int64)
        for y in range(oh):
            yin[y] = -1 if yo < 0.0 else int(yo)
            yo += m[4]
        okx = (xin >= 0) & (xin < w)
        oky = (yin >= 0) & (yin < h)
        if okx.any() and oky.any():
            xs = np.nonzero(okx)[0]
            xmin, xmax = xs[0], xs[-1] + 1  # the C loop copies the whole [xmin,xmax) span
            ys = np.nonzero(oky)[0]
            span = np.arange(xmin, xmax)
            sub = a[yin[ys]][:, np.clip(xin[span], 0, w - 1)]
            out[np.ix_(ys, span)] = sub
        return out[:, :, 0] if was2d else out
    a0, a1, a3, a4 = (_fix16(m[0]), _fix16(m[1]), _fix16(m[3]), _fix16(m[4]))
    a2 = _fix16(m[2] + 0.5 * m[0] + 0.5 * m[1])
    a5 = _fix16(m[5] + 0.5 * m[3] + 0.5 * m[4])
    x = np.arange(ow, dtype=np.int64)[None, :]
    y = np.arange(oh, dtype=np.int64)[:, None]
    xin = (a2 + a1 * y + a0 * x) >> 16
    yin = (a5 + a4 * y + a3 * x) >> 16
    ok = (xin >= 0) & (xin < w) & (yin >= 0) & (yin < h)
    src = a[np.clip(yin, 0, h - 1), np.clip(xin, 0, w - 1)]
    out = np.where(ok[:, :, None], src, out)
    return out[:, :, 0] if was2d else out


def _affine_coords(oh, ow, m):
    x = np.arange(ow, dtype=np.float64)[None, :] + 0.5
    y = np.arange(oh, dtype=np.float64)[:, None] + 0.5
    xin = m[0] * x + m[1] * y + m[2]
    yin = m[3] * x + m[4] * y + m[5]
    return xin, yin


def affine_bilinear(img, out_size, m, fill=None, return_float=False):
    """Image.transform(size, AFFINE, m, BILINEAR, fillcolor=fill): float64
    coordinates and lerps, neighbour clamp, `(UINT8)v` truncation."""
    a, was2d = _as3d(img)
    h, w, c = a.shape
    ow, oh = out_size
    xin, yin = _affine_coords(oh, ow, m)
    ok = (xin >= 0.0) & (xin < w) & (yin >= 0.0) & (yin < h)
    xf = xin - 0.5
    yf = yin - 0.5
    x0 = np.floor(xf)
    y0 = np.floor(yf)
    dx = (xf - x0)[:, :, None]
    dy = (yf - y0)[:, :, None]
    x0 = x0.astype(np.int64)
    y0 = y0.astype(np.int64)
    xa = np.clip(x0, 0, w - 1)
    xb = np.clip(x0 + 1, 0, w - 1)
    ya = np.clip(y0, 0, h - 1)
    src = a.astype(np.float64)
    p00, p01 = src[ya, xa], src[ya, xb]
    v1 = p00 + (p01 - p00) * dx
    has_row2 = ((y0 + 1 >= 0) & (y0 + 1 < h))[:, :, None]
    yb = np.clip(y0 + 1, 0, h - 1)
    p10, p11 = src[yb, xa], src[yb, xb]
    v2 = np.where(has_row2, p10 + (p11 - p10) * dx, v1)
    v = v1 + (v2 - v1) * dy
    if return_float:
        fillv = _fill_array(oh, ow, c, fill, np.float64)
        outf = np.where(ok[:, :, None], v, fillv)
        return (outf[:, :, 0] if was2d else outf), ok
    out = _fill_array(oh, ow, c, fill)
    out = np.where(ok[:, :, None], v.astype(np.int64).astype(np.uint8), out)
    return out[:, :, 0] if was2d else out


def affine_bicubic(img, out_size, m, fill=None, return_float=False):
    """Image.transform(size, AFFINE, m, BICUBIC, fillcolor=fill)
    (used by apply_shear, transformation.py:217-224)."""
    a, was2d = _as3d(img)
    h, w, c = a.shape
    ow, oh = out_size
    xin, yin = _affine_coords(oh, ow, m)
    ok = (xin >= 0.0) & (xin < w) & (yin >= 0.0) & (yin < h)
    xf = xin - 0.5
    yf = yin - 0.5
    x0 = np.floor(xf)
    y0 = np.floor(yf)
    dx = (xf - x0)[:, :, None]
    dy = (yf - y0)[:, :, None]
    x0 = x0.astype(np.int64) - 1
    y0 = y0.astype(np.int64) - 1
    src = a.astype(np.float64)

    def cubic(v1, v2, v3, v4, d):
        p1 = v2
        p2 = -v1 + v3
        p3 = 2 * (v1 - v2) + v3 - v4
        p4 = -v1 + v2 - v3 + v4
        return p1 + d * (p2 + d * (p3 + d * p4))

    xs = [np.clip(x0 + k, 0, w - 1) for k in range(4)]

    def row(yy):
        yc = np.clip(yy, 0, h - 1)
        return cubic(src[yc, xs[0]], src[yc, xs[1]], src[yc, xs[2]], src[yc, xs[3]], dx)

    v1 = row(y0)
    in1 = ((y0 + 1 >= 0) & (y0 + 1 < h))[:, :, None]
    in2 = ((y0 + 2 >= 0) & (y0 + 2 < h))[:, :, None]
    in3 = ((y0 + 3 >= 0) & (y0 + 3 < h))[:, :, None]
    v2 = np.where(in1, row(y0 + 1), v1)
    v3 = np.where(in2, row(y0 + 2), v2)
    v4 = np.where(in3, row(y0 + 3), v3)
    v = cubic(v1, v2, v3, v4, dy)
    if return_float:
        fillv = _fill_array(oh, ow, c, fill, np.float64)
        outf = np.where(ok[:, :, None], v, fillv)
        return (outf[:, :, 0] if was2d else outf), ok
    q = np.where(v <= 0.0, 0, np.where(v >= 255.0, 255, v.astype(np.int64))).astype(np.uint8)
    out = _fill_array(oh, ow, c, fill)
    out = np.where(ok[:, :, None], q, out)
    return out[:, :, 0] if was2d else out


def apply_rotation(img, angle):
    """transformation.py:198-201: img.rotate(-angle, fillcolor=(0,0,0), expand=False),
    default resample NEAREST."""
    a = np.asarray(img)
    h, w = a.shape[:2]
    kind, m = rotate_plan(w, h, -angle)
    if kind == "copy":
        return a.copy()
    if kind == "rot180":
        return np.ascontiguousarray(a[::-1, ::-1])
    if kind == "rot90":   # Transpose.ROTATE_90 = counter-clockwise
        return np.ascontiguousarray(np.rot90(a, 1))
    if kind == "rot270":
        return np.ascontiguousarray(np.rot90(a, 3))
    return affine_nearest(a, (w, h), m, fill=(0, 0, 0))


def rotate_bilinear(img, angle):
    """img.rotate(angle, resample=BILINEAR, fillcolor=0) — the benchmark variant."""
    a = np.asarray(img)
    h, w = a.shape[:2]
    kind, m = rotate_plan(w, h, angle)
    if kind != "affine":
        return apply_rotation(a, -angle)
    return affine_bilinear(a, (w, h), m, fill=(0, 0, 0))


def shear_geometry(w, h, shear_factor):
    """transformation.py:213-221 -> (new_width, matrix)."""
    shift = int(math.ceil(shear_factor * h))
    m = (1, shear_factor, -shift if shear_factor > 0 else 0, 0, 1, 0)
    return w + shift, m


def apply_shear(img, shear_factor):
    """transformation.py:212-226: BICUBIC affine, white fill, widened output."""
    a = np.asarray(img)
    h, w = a.shape[:2]
    nw, m = shear_geometry(w, h, shear_factor)
    return affine_bicubic(a, (nw, h), [float(v) for v in m], fill=(255, 255, 255))


# ----------------------------------------------------------------------------
# a3: Lanczos resize (libImaging Resample.c), apply_scale
# ----------------------------------------------------------------------------

PRECISION_BITS = 32 - 8 - 2


def _lanczos3(x):
    if -3.0 <= x < 3.0:
        def sinc(t):
            if t == 0.0:
                return 1.0
            t = t * math.pi
            return math.sin(t) / t
        return sinc(x) * sinc(x / 3)
    return 0.0


# the other convolution filters of Resample.c; keys are Pillow's Image.Resampling values
RESAMPLE_LANCZOS, RESAMPLE_BILINEAR, RESAMPLE_BICUBIC, RESAMPLE_BOX, RESAMPLE_HAMMING = 1, 2, 3, 4, 5


def _box_filter(x):
    return 1.0 if -0.5 < x <= 0.5 else 0.0


def _bilinear_filter(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def _hamming_filter(x):
    x = abs(x)
    if x == 0.0:
        return 1.0
    if x >= 1.0:
        return 0.0
    x = x * math.pi
    # Resample.c writes the window constants as float literals: 0.54f + 0.46f * cos(x)
    return math.sin(x) / x * (float(np.float32(0.54)) + float(np.float32(0.46)) * math.cos(x))


def _bicubic_filter(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


_FILTERS = {RESAMPLE_LANCZOS: (_lanczos3, 3.0), RESAMPLE_BILINEAR: (_bilinear_filter, 1.0),
            RESAMPLE_BICUBIC: (_bicubic_filter, 2.0), RESAMPLE_BOX: (_box_filter, 0.5),
            RESAMPLE_HAMMING: (_hamming_filter, 1.0)}


def lanczos_coeffs(in_size, out_size, resample=RESAMPLE_LANCZOS):
    """precompute_coeffs + normalize_coeffs_8bpc for the whole-image box.
    Returns (bounds[out,2] int32 (xmin, count), kk[out,ksize] int32, ksize)."""
    filt, fsupport = _FILTERS[resample]
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ws = [filt((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in ws:
            ww += v
        for x in range(xmax):
            v = ws[x] / ww if ww != 0.0 else ws[x]
            if v < 0:
                kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS))
            else:
                kk[xx, x] = int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _resample_axis0(a, out_size, resample=RESAMPLE_LANCZOS):
    """Resample along axis 0 of an (n, m, c) uint8 array."""
    n = a.shape[0]
    bounds, kk, ksize = lanczos_coeffs(n, out_size, resample)
    src = a.astype(np.int64)
    acc = np.full((out_size,) + a.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
    idx = bounds[:, 0][:, None] + np.arange(ksize)[None, :]
    idx = np.minimum(idx, n - 1)  # coefficients beyond count are zero
    for t in range(ksize):
        acc += src[idx[:, t]] * kk[:, t].astype(np.int64)[:, None, None]
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_lanczos(img, size):
    """img.resize(size, LANCZOS) — transformation.py:179."""
    return resize(img, size, RESAMPLE_LANCZOS)


def resize(img, size, resample=RESAMPLE_BICUBIC):
    """img.resize(size, resample): horizontal pass, uint8 intermediate, vertical pass;
    a pass is skipped when that dimension is unchanged (Resample.c ImagingResample)."""
    a, was2d = _as3d(img)
    h, w, _ = a.shape
    nw, nh = size
    out = a
    if nw != w:
        out = np.ascontiguousarray(_resample_axis0(out.transpose(1, 0, 2), nw, resample).transpose(1, 0, 2))
    if nh != h:
        out = _resample_axis0(out, nh, resample)
    out = np.ascontiguousarray(out) if out is not a else a.copy()
    return out[:, :, 0] if was2d else out


def scale_geometry(w, h, scale_factor):
    """transformation.py:174-194 -> (nw, nh, mode, ox, oy) with mode in crop|paste|none."""
    nw = int(w * scale_factor)
    nh = int(h * scale_factor)
    if scale_factor > 1.0:
        return nw, nh, "crop", (nw - w) // 2, (nh - h) // 2
    if scale_factor < 1.0:
        return nw, nh, "paste", (w - nw) // 2, (h - nh) // 2
    return nw, nh, "none", 0, 0


def apply_scale(img, scale_factor):
    """transformation.py:173-196."""
    a = np.asarray(img)
    h, w = a.shape[:2]
    nw, nh, mode, ox, oy = scale_geometry(w, h, scale_factor)
    scaled = resize_lanczos(a, (nw, nh))
    if mode == "crop":
        return np.ascontiguousarray(scaled[oy:oy + h, ox:ox + w])
    if mode == "paste":
        out = np.zeros_like(a)
        out[oy:oy + nh, ox:ox + nw] = scaled
        return out
    return scaled


def apply_camera_distance(img, distance_factor):
    """transformation.py:309-314."""
    return apply_scale(img, 2.75 / distance_factor)


# ----------------------------------------------------------------------------
# translation (transformation.py:284-307)
# ----------------------------------------------------------------------------

def apply_translation(img, tx, ty):
    a = np.asarray(img)
    h, w = a.shape[:2]
    out = np.zeros_like(a)
    px, py = int(tx), int(ty)
    cl, ct = max(0, -px), max(0, -py)
    cr, cb = min(w, w - px), min(h, h - py)
    if cl < cr and ct < cb:
        rx, ry = max(0, px), max(0, py)
        out[ry:ry + (cb - ct), rx:rx + (cr - cl)] = a[ct:cb, cl:cr]
    return out


def apply_xy_translation_3d(img, tx, ty):
    """transformation.py:316-321."""
    a = np.asarray(img)
    h, w = a.shape[:2]
    return apply_translation(a, int(tx * w), int(ty * h))


# ----------------------------------------------------------------------------
# a4: Sobel + background change (transformation.py:328-345)
# ----------------------------------------------------------------------------

def sobel_scipy(gray, axis=-1):
    """scipy.ndimage.sobel(uint8 array, axis) with uint8 output: derivative
    [-1,0,1] along `axis`, smoothing [1,2,1] along the other, mode 'reflect',
    result stored mod 256."""
    g = np.asarray(gray).astype(np.int64)
    h, w = g.shape
    axis = axis % 2
    ys, xs = np.arange(h), np.arange(w)

    def shift(arr, d, ax):
        if ax == 0:
            return arr[reflect_index(ys + d, h), :]
        return arr[:, reflect_index(xs + d, w)]

    d = shift(g, 1, axis) - shift(g, -1, axis)
    o = 1 - axis
    s = shift(d, -1, o) + 2 * d + shift(d, 1, o)
    return np.mod(s, 256).astype(np.uint8)


def sobel_gradients(gray):
    """Exact integer Gx, Gy (no wrap) with SciPy 'reflect' borders."""
    g = np.asarray(gray).astype(np.int64)
    h, w = g.shape
    ys, xs = np.arange(h), np.arange(w)
    up, dn = g[reflect_index(ys - 1, h), :], g[reflect_index(ys + 1, h), :]
    lf, rt = g[:, reflect_index(xs - 1, w)], g[:, reflect_index(xs + 1, w)]
    dxv = rt - lf
    gx = dxv[reflect_index(ys - 1, h), :] + 2 * dxv + dxv[reflect_index(ys + 1, h), :]
    dyv = dn - up
    gy = dyv[:, reflect_index(xs - 1, w)] + 2 * dyv + dyv[:, reflect_index(xs + 1, w)]
    return gx, gy


def sobel_magnitude_f(gray):
    gx, gy = sobel_gradients(gray)
    return np.sqrt((gx * gx + gy * gy).astype(np.float64))


def sobel_magnitude(gray):
    """Benchmark configs[2]: sqrt(Gx^2+Gy^2) on the L image, saturate-round to uint8.
    No reference counterpart; defined by this oracle."""
    return saturate_u8(sobel_magnitude_f(gray))


def rgb_sobel_magnitude(img):
    return sobel_magnitude(rgb2l(img))


def percentile_linear_u8(values_u8, q):
    """np.percentile(uint8 array, q) (default 'linear' method) from a histogram."""
    v = np.asarray(values_u8).ravel()
    n = v.size
    hist = np.bincount(v, minlength=256)
    cum = np.cumsum(hist)
    quant = q / 100.0
    # numpy _compute_virtual_index with alpha = beta = 1
    virt = n * quant + (1.0 + quant * (1.0 - 1.0 - 1.0)) - 1.0
    lo = int(math.floor(virt))
    gamma = virt - lo
    lo = min(max(lo, 0), n - 1)
    hi = min(lo + 1, n - 1)
    a = float(np.searchsorted(cum, lo, side="right"))
    b = float(np.searchsorted(cum, hi, side="right"))
    # numpy _lerp
    diff = b - a
    out = a + diff * gamma
    if gamma >= 0.5:
        out = b - diff * (1.0 - gamma)
    if diff == 0:
        out = a
    return out


def binary_dilation_cross(mask, iterations=1):
    """scipy.ndimage.binary_dilation(mask, iterations=k): 4-connected cross,
    border value 0."""
    m = np.asarray(mask, bool)
    for _ in range(iterations):
        p = np.pad(m, 1)
        m = p[1:-1, 1:-1] | p[:-2, 1:-1] | p[2:, 1:-1] | p[1:-1, :-2] | p[1:-1, 2:]
    return m


def composite(img1, img2, mask_u8):
    """Image.composite(img1, img2, mask) for an L mask holding only 0/255."""
    m = np.asarray(mask_u8) != 0
    return np.where(m[:, :, None], np.asarray(img1), np.asarray(img2))


def apply_background_change(img, bg_color):
    """transformation.py:328-345."""
    a = np.asarray(img)
    if a.shape[2] == 4:
        a = a[:, :, :3]
    bg = np.empty_like(a)
    bg[...] = np.array([int(c * 255) for c in bg_color], np.uint8)
    gray = rgb2l(a)
    edges = sobel_scipy(gray)
    thr = percentile_linear_u8(edges, 70)
    fg = binary_dilation_cross(edges > thr, 3)
    return composite(a, bg, (fg * 255).astype(np.uint8))


# ----------------------------------------------------------------------------
# AugMix operation set (fall_2025/AugMix.py:30-37) and the Shannon-entropy feature
# (fall_2025/Initial_Experiments.py:95-113).  Pinned against PIL.ImageOps / np.histogram +
# scipy.stats.entropy in tests/test_oracle_vs_libs.py.
# ----------------------------------------------------------------------------

# ---- TransformationPool.histogram_equalization (cifar_image_transformations.py:122-129) -------
# PARITY UNPINNED: cv2 is not installed; OpenCV's 8-bit integer definitions restated
# (imgproc color_yuv.simd.hpp RGB2YCrCb_i<uchar> / YCrCb2RGB_i<uchar> with the YUV coefficient
# sets, yuv_shift = 14; histogram.cpp equalizeHist).

def _descale14(x):
    return (x + (1 << 13)) >> 14          # CV_DESCALE, arithmetic shift


def rgb2yuv_cv(img):
    a = np.asarray(img, np.uint8).astype(np.int64)
    R, G, B = a[..., 0], a[..., 1], a[..., 2]
    Y = _descale14(R * 4899 + G * 9617 + B * 1868)
    V = _descale14((R - Y) * 14369 + (128 << 14))
    U = _descale14((B - Y) * 8061 + (128 << 14))
    return np.clip(np.stack([Y, U, V], axis=-1), 0, 255).astype(np.uint8)


def yuv2rgb_cv(img):
    a = np.asarray(img, np.uint8).astype(np.int64)
    Y, U, V = a[..., 0], a[..., 1] - 128, a[..., 2] - 128
    b = Y + _descale14(U * 33292)
    g = Y + _descale14(U * -6472 + V * -9519)
    r = Y + _descale14(V * 18678)
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)


def equalize_hist_cv(gray):
    """cv2.equalizeHist of one 8-bit plane."""
    g = np.asarray(gray, np.uint8)
    hist = np.bincount(g.ravel(), minlength=256)
    total = int(g.size)
    i = int(np.nonzero(hist)[0][0])
    if hist[i] == total:
        return np.full_like(g, i)
    scale = np.float32(255.0) / np.float32(total - int(hist[i]))
    lut = np.zeros(256, np.uint8)
    acc = 0
    for k in range(i + 1, 256):
        acc += int(hist[k])
        lut[k] = np.clip(np.rint(np.float32(acc) * scale), 0, 255)     # saturate_cast<uchar>(float): round half even
    return lut[g]


def histogram_equalization(img):
    yuv = rgb2yuv_cv(img)
    yuv[..., 0] = equalize_hist_cv(yuv[..., 0])
    return yuv2rgb_cv(yuv)


def vert_flip(img):
    """fall_2025/transformations_code:39-41: img.transpose(FLIP_LEFT_RIGHT)."""
    return np.ascontiguousarray(np.asarray(img)[:, ::-1])


def rand_crop(img, x, y):
    """fall_2025/transformations_code:43-48 for the drawn corner (x, y):
    img.crop((x, y, x+cs, y+cs)).resize((32, 32)) with cs = int(0.78 * w), BICUBIC default."""
    a = np.asarray(img, np.uint8)
    cs = int(0.78 * a.shape[1])
    return resize(np.ascontiguousarray(a[y:y + cs, x:x + cs]), (32, 32), RESAMPLE_BICUBIC)


def _fma32(a, b, c):
    """Exact fp32 fused multiply-add.  The product of two fp32 values is exact in fp64; the sum
    with c is rounded to odd in fp64 (TwoSum gives the rounding error, an inexact even result
    moves to its odd neighbour on the error's side), after which the rounding to fp32 is the
    single correct one (53 >= 2*24 + 2)."""
    p = np.asarray(a, np.float64) * np.asarray(b, np.float64)
    c = np.asarray(c, np.float64)
    p, c = np.broadcast_arrays(p, c)
    s = p + c
    bb = s - p
    e = (p - (s - bb)) + (c - bb)
    even = (s.view(np.int64) & 1) == 0
    nudge = (e != 0) & even & np.isfinite(s)
    s = np.where(nudge, np.nextafter(s, np.where(e > 0, np.inf, -np.inf)), s)
    return s.astype(np.float32)


def perspective_endpoints(width, height, distortion_scale, randint):
    """torchvision.transforms.RandomPerspective.get_params (torchvision/transforms/transforms.py),
    the draw order behind fall_2025/transformations_code:61-65: `randint(lo, hi)` stands for
    int(torch.randint(lo, hi, size=(1,)).item()), called eight times in this order."""
    hh, hw = height // 2, width // 2
    dw, dh = int(distortion_scale * hw), int(distortion_scale * hh)
    topleft = [randint(0, dw + 1), randint(0, dh + 1)]
    topright = [randint(width - dw - 1, width), randint(0, dh + 1)]
    botright = [randint(width - dw - 1, width), randint(height - dh - 1, height)]
    botleft = [randint(0, dw + 1), randint(height - dh - 1, height)]
    start = [[0, 0], [width - 1, 0], [width - 1, height - 1], [0, height - 1]]
    return start, [topleft, topright, botright, botleft]


def perspective_coeffs(startpoints, endpoints):
    """torchvision.transforms.functional._get_perspective_coeffs: the eight coefficients that map
    an OUTPUT pixel to its source, least squares in fp64 (here an 8x8 solve), cast to fp32."""
    a = np.zeros((8, 8), np.float64)
    for i, (p1, p2) in enumerate(zip(endpoints, startpoints)):
        a[2 * i] = [p1[0], p1[1], 1, 0, 0, 0, -p2[0] * p1[0], -p2[0] * p1[1]]
        a[2 * i + 1] = [0, 0, 0, p1[0], p1[1], 1, -p2[1] * p1[0], -p2[1] * p1[1]]
    b = np.asarray(startpoints, np.float64).reshape(8)
    return np.linalg.solve(a, b).astype(np.float32)


def perspective_grid(coeffs, ow, oh):
    """torchvision.transforms._functional_tensor._perspective_grid in fp32: normalised sampling
    coordinates for pixel centres (x+0.5, y+0.5).  The two `bmm`s reduce over three terms; the
    accumulation order fma(1, c, fma(y, b, x*a)) is the one the installed torch 2.10 CPU build
    produces bit-for-bit (tests/test_oracle_vs_libs.py)."""
    f32 = np.float32
    c = [f32(v) for v in coeffs]
    sx, sy = f32(0.5 * ow), f32(0.5 * oh)
    t = [c[0] / sx, c[1] / sx, c[2] / sx, c[3] / sy, c[4] / sy, c[5] / sy]
    x = (np.arange(ow, dtype=f32) + f32(0.5))[None, :]
    y = (np.arange(oh, dtype=f32) + f32(0.5))[:, None]

    def dot3(a, b, cc):
        return _fma32(y, b, x * a) + cc

    den = dot3(c[6], c[7], f32(1))
    return dot3(t[0], t[1], t[2]) / den - f32(1), dot3(t[3], t[4], t[5]) / den - f32(1)


def grid_sample_bilinear_zeros(t, gx, gy):
    """torch.nn.functional.grid_sample(mode='bilinear', padding_mode='zeros', align_corners=False)
    on an HWC fp32 image (ATen/native/cpu/GridSamplerKernel.cpp), with the contraction pattern of
    the installed CPU build: unnormalise = fma(g+1, size, -1)/2; weights from the distances to the
    four sides; the four taps accumulated nw, ne, sw, se as one multiply then three fmas.
    Returns (sampled image, sampled all-ones mask)."""
    f32 = np.float32
    h, w = t.shape[:2]
    one = f32(1)
    ix = _fma32(gx + one, f32(w), f32(-1)) / f32(2)
    iy = _fma32(gy + one, f32(h), f32(-1)) / f32(2)
    x0, y0 = np.floor(ix), np.floor(iy)
    ww = ix - x0
    we = one - ww
    wn = iy - y0
    ws = one - wn
    wts = [ws * we, ws * ww, wn * we, wn * ww]
    xi, yi = x0.astype(np.int64), y0.astype(np.int64)
    acc = msk = None
    for (dx, dy), wt in zip(((0, 0), (1, 0), (0, 1), (1, 1)), wts):
        xx, yy = xi + dx, yi + dy
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        v = np.where(ok[..., None], t[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)], f32(0))
        m = np.where(ok, one, f32(0))
        if acc is None:
            acc, msk = v * wt[..., None], m * wt
        else:
            acc = _fma32(v, np.broadcast_to(wt[..., None], v.shape), acc)
            msk = _fma32(m, wt, msk)
    return acc, msk


def perspective_warp(img, coeffs):
    """fall_2025/transformations_code:54-66 for drawn coefficients: ToTensor (u8/255 in fp32),
    F.perspective(BILINEAR, fill=[0,0,0]) = grid_sample of the image plus a ones channel, then
    img*mask + (1-mask)*fill (torchvision _apply_grid_transform), ToPILImage = mul(255).byte()
    (truncation)."""
    a = np.asarray(img, np.uint8)
    if a.ndim == 2:
        a = a[..., None]
    h, w = a.shape[:2]
    f32 = np.float32
    t = a.astype(f32) / f32(255)
    gx, gy = perspective_grid(coeffs, w, h)
    s, m = grid_sample_bilinear_zeros(t, gx, gy)
    m = m[..., None]
    out = (s * m + (f32(1) - m) * f32(0)) * f32(255)
    out = np.minimum(out, f32(255)).astype(np.uint8)
    return out if np.asarray(img).ndim == 3 else out[..., 0]


def posterize_lut(bits):
    """ImageOps.posterize (PIL/ImageOps.py): lut[i] = i & ~(2**(8-bits)-1)."""
    mask = ~(2 ** (8 - int(bits)) - 1)
    return np.array([i & mask for i in range(256)], np.uint8)


def solarize_lut(threshold=128):
    """ImageOps.solarize: i below the threshold, 255 - i from it on."""
    return np.array([i if i < threshold else 255 - i for i in range(256)], np.uint8)


def apply_lut(img, lut):
    """Image.point(lut) with one 256-entry table per channel (lut: [256] or [c,256])."""
    img = np.asarray(img, np.uint8)
    lut = np.asarray(lut, np.uint8)
    if img.ndim == 2:
        return lut.reshape(-1, 256)[0][img]
    if lut.ndim == 1:
        return lut[img]
    return np.stack([lut[c][img[..., c]] for c in range(img.shape[-1])], axis=-1)


def channel_histogram(img):
    """[c,256] int64 histogram of an HWC / HW uint8 image."""
    img = np.asarray(img, np.uint8)
    if img.ndim == 2:
        img = img[..., None]
    return np.stack([np.bincount(img[..., c].ravel(), minlength=256) for c in range(img.shape[-1])])


def equalize_lut(h):
    """ImageOps.equalize's table for one 256-bin histogram (PIL/ImageOps.py: equalize);
    Image.point clips entries to 0..255 (_imaging.c getlist, CLIP8)."""
    h = [int(v) for v in h]
    histo = [v for v in h if v]
    if len(histo) <= 1:
        return np.arange(256, dtype=np.uint8)
    step = (sum(histo) - histo[-1]) // 255
    if not step:
        return np.arange(256, dtype=np.uint8)
    lut = []
    n = step // 2
    for i in range(256):
        lut.append(min(n // step, 255))
        n += h[i]
    return np.array(lut, np.uint8)


def equalize(img):
    """ImageOps.equalize(img) (fall_2025/AugMix.py:36)."""
    img = np.asarray(img, np.uint8)
    hist = channel_histogram(img)
    return apply_lut(img, np.stack([equalize_lut(hist[c]) for c in range(hist.shape[0])]))


def posterize(img, bits):
    return apply_lut(img, posterize_lut(bits))


def solarize(img, threshold=128):
    return apply_lut(img, solarize_lut(threshold))


def shannon_entropy_from_histogram(counts):
    """compute_shannon_entropy (Initial_Experiments.py:95-113) for an image whose float values
    are v/255: np.histogram(x, bins=256, range=(0,1), density=True) puts byte v in bin v
    (255 -> the closed last bin), so the density is counts / N / (1/256); scipy.stats.entropy
    normalises, takes -sum(p log p) and divides by log(2)."""
    counts = np.asarray(counts, np.float64).ravel()
    edges = np.linspace(0.0, 1.0, 257)
    dens = counts / np.diff(edges) / counts.sum()
    dens = dens[dens > 0]
    pk = dens / np.sum(dens)
    return float(np.sum(-pk * np.log(pk)) / np.log(2.0))


# ----------------------------------------------------------------------------
# benchmark configs[0]: grayscale + 3x3 box blur
# ----------------------------------------------------------------------------

def gray_box3(img):
    return conv2d(rgb2l(img), box_kernel(3))


# ----------------------------------------------------------------------------
# H: driver value grids (transformation.py:95-105,125-139)
# ----------------------------------------------------------------------------

TRANSFORM_GRID = {
    "scale": (0.9, 1.4, 0.1),
    "rotation": (-22.5, 22.5, 2.5),
    "lighten_darken": (-0.05, 0.05, 0.01),
    "gaussian_noise": (0.0, 0.1, 0.01),
    "translation": (-50, 50, 5),
    "contrast": (0, 1, 0.1),
    "blur": (0, 5, 0.5),
    "shear": (0, 1, 0.1),
}


def grid_values(name):
    lo, hi, step = TRANSFORM_GRID[name]
    num_steps = int((hi - lo) / step) + 1
    return [lo + j * step for j in range(num_steps)]
