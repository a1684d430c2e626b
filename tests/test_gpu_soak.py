"""Soak for the kernels added in round 2 (fused matrix-core resample, matrix-core Gaussians in float and
fixed-point mode, separable filter2D routing, translation / mirrors / quarter turns, wide-lane 3x3 and box
filters): IMGXF_SOAK=<n> runs n further seeds of random geometries against the oracle / Pillow; the default
suite runs 2 seeds.  Every check is bit-exact except the float filters (fp64 oracle, tie tolerance)."""
import os

import numpy as np
import pytest
import torch
from PIL import Image, ImageFilter

from oracle import imgxf_oracle as O
from test_gpu_parity import MFMA_ABS, assert_quantised_close

pytestmark = pytest.mark.gpu
SEEDS = int(os.environ.get("IMGXF_SOAK", "2"))


def _img(rng, h, w, c=3):
    kind = int(rng.integers(0, 3))
    if kind == 0:
        return rng.integers(0, 256, (h, w, c), dtype=np.uint8)
    if kind == 1:
        y, x = np.mgrid[0:h, 0:w]
        return np.stack([((x * 255 // max(w - 1, 1) + y * 3 + 40 * k) % 256) for k in range(c)], -1).astype(np.uint8)
    a = np.full((h, w, c), int(rng.integers(0, 256)), np.uint8)
    a[rng.random((h, w)) < 0.03] = rng.integers(0, 256, c)
    return a


@pytest.mark.parametrize("seed", range(SEEDS))
def test_soak_round2_kernels(device, seed):
    from imagetransformations_amd import ops
    from imagetransformations_amd import transformation as T
    rng = np.random.default_rng(770000 + seed)
    h, w = int(rng.integers(33, 240)), int(rng.integers(6, 40)) * 16
    a = _img(rng, h, w)
    t = torch.from_numpy(a).to(device)
    # fused resample: any filter, both directions, optional crop window
    sx, sy = float(rng.uniform(0.62, 2.2)), float(rng.uniform(0.62, 2.2))
    nw, nh = max(1, int(w * sx)), max(1, int(h * sy))
    flt = int(rng.choice([1, 1, 2, 3, 4, 5]))
    ref = O.resize(a, (nw, nh), flt)
    assert np.array_equal(ops.resize(t, (nw, nh), flt).cpu().numpy(), ref), ("resize", h, w, nw, nh, flt)
    if nw > 8 and nh > 8 and nw != w and nh != h:
        l, tp = int(rng.integers(0, nw // 2)), int(rng.integers(0, nh // 2))
        r, b = int(rng.integers(l + 1, nw + 1)), int(rng.integers(tp + 1, nh + 1))
        assert np.array_equal(ops.resize_crop(t, (nw, nh), (l, tp, r, b), flt).cpu().numpy(), ref[tp:b, l:r]), ("crop", l, tp, r, b)
    s = float(rng.choice(O.grid_values("scale")))
    assert np.array_equal(T._scale_t(t, s).cpu().numpy(), O.apply_scale(a, s)), ("apply_scale", h, w, s)
    # large-radius Gaussian: float definition (tolerance) and fixed-point mode (bit-exact)
    radius = float(rng.choice([2.0, 2.5, 3.0, 3.5, 4.0, 4.5, 5.0]))
    k = O.blur_ksize(radius)
    out, f32 = ops.gaussian_blur(t, k, radius, return_f32=True)
    assert_quantised_close(out.cpu().numpy(), f32.cpu().numpy(), O.gaussian_blur_f64(a, k, radius), O.saturate_u8, abs_term=MFMA_ABS * float(a.max()), max_tie_fraction=2e-2)
    assert np.array_equal(ops.gaussian_blur(t, k, radius, fixed_point=True).cpu().numpy(), O.gaussian_blur_cv_fixed(a, k, radius)), ("fixed", k)
    # filter2D: motion blur rows and a random outer product
    size = int(rng.choice([5, 7, 9, 11, 13, 15]))
    for kern in (O.motion_blur_kernel(size), np.outer(rng.uniform(0, 0.4, int(rng.choice([1, 3, 5, 9]))), rng.uniform(0, 0.4, int(rng.choice([3, 7, 11]))))):
        ref_f = O.conv2d_f64(a, kern)
        got = ops.conv2d(t, np.asarray(kern).tolist()).cpu().numpy()
        d = np.abs(got.astype(int) - O.saturate_u8(ref_f).astype(int))
        tie = np.abs(ref_f - np.floor(ref_f) - 0.5) < 1e-4
        assert d.max() <= 1 and (d == 0)[~tie].all(), ("filter2D", kern.shape)
    # integer geometry
    tx, ty = float(rng.uniform(-1.2 * w, 1.2 * w)), float(rng.uniform(-1.2 * h, 1.2 * h))
    assert np.array_equal(T._translation_t(t, tx, ty).cpu().numpy(), O.apply_translation(a, tx, ty)), ("translation", tx, ty)
    assert np.array_equal(ops.flip(t).cpu().numpy(), a[:, ::-1]) and np.array_equal(ops.flip(t, True).cpu().numpy(), a[::-1])
    for q in (1, 2, 3):
        assert np.array_equal(ops.rot90(t, q).cpu().numpy(), np.rot90(a, q)), ("rot90", q)
    # Pillow's own filters
    img = Image.fromarray(a)
    assert np.array_equal(ops.filter3x3(t, ops.SMOOTH_KERNEL, 13).cpu().numpy(), np.asarray(img.filter(ImageFilter.SMOOTH)))
    rad = float(rng.choice([0.5, 1.0, 2.0, 3.0, 4.0, 5.0, 6.5]))
    assert np.array_equal(ops.gaussian_blur_pil(t, rad).cpu().numpy(), np.asarray(img.filter(ImageFilter.GaussianBlur(rad)))), ("defocus", rad)
    brad = float(rng.uniform(0.2, 5.0))
    assert np.array_equal(ops.box_blur(t, brad).cpu().numpy(), np.asarray(img.filter(ImageFilter.BoxBlur(brad)))), ("box", brad)


@pytest.mark.parametrize("seed", range(SEEDS))
def test_soak_batched_kernels(device, seed):
    """Batches: frames-in-the-workgroup bilinear (precise, bit-exact with Pillow's arithmetic), nearest rotation,
    super-row Gaussian strips (small radii), shear — random frame counts, matrices and sizes."""
    from imagetransformations_amd import ops
    rng = np.random.default_rng(880000 + seed)
    n = int(rng.integers(2, 8))
    h, w = int(rng.integers(40, 200)), int(rng.integers(6, 30)) * 16
    a = np.stack([_img(rng, h, w) for _ in range(n)])
    t = torch.from_numpy(a).to(device)
    m = O.rotate_zoom_matrix(w, h, float(rng.uniform(-180, 180)), float(rng.uniform(0.6, 2.2)))
    fill = tuple(int(v) for v in rng.integers(0, 256, 3))
    got = ops.affine(t, m, (w, h), ops.BILINEAR, fill, precise=True).cpu().numpy()
    for i in range(n):
        assert np.array_equal(got[i], O.affine_bilinear(a[i], (w, h), m, fill=fill)), ("bilinear", n, h, w, m, i)
    ang = float(rng.choice(O.grid_values("rotation") + [float(rng.uniform(-180, 180))]))
    got = ops.rotate(t, -ang, ops.NEAREST, (0, 0, 0)).cpu().numpy()
    for i in range(n):
        assert np.array_equal(got[i], O.apply_rotation(a[i], ang)), ("nearest", ang, i)
    radius = float(rng.choice([0.5, 5 / 6, 1.0, 1.5]))
    k = O.blur_ksize(radius)
    out, f32 = ops.gaussian_blur(t, k, radius, return_f32=True)
    out, f32 = out.cpu().numpy(), f32.cpu().numpy()
    for i in range(n):
        assert_quantised_close(out[i], f32[i], O.gaussian_blur_f64(a[i], k, radius), O.saturate_u8, max_tie_fraction=2e-2)   # pure ramps: many ties
    sh = float(rng.choice(O.grid_values("shear")[1:]))
    nw, ms = O.shear_geometry(w, h, sh)
    got = ops.affine(t, ms, (nw, h), ops.BICUBIC, (255, 255, 255), precise=True).cpu().numpy()
    for i in (0, n - 1):
        assert np.array_equal(got[i], O.apply_shear(a[i], sh)), ("shear", sh, i)


@pytest.mark.parametrize("seed", range(max(1, SEEDS // 100)))
def test_soak_full_size_equivalences(device, seed, monkeypatch):
    """4K / 1080p frames: the matrix-core kernels against the kernels they replaced (no oracle at this size):
    fused resample == two-pass, fixed-point Gaussian on the i8 cores == vector kernels, float Gaussian within a
    rounding tie of the vector kernels, translation == fill + paste."""
    from imagetransformations_amd import ops
    from imagetransformations_amd import transformation as T
    rng = np.random.default_rng(990000 + seed)
    h, w = ((2160, 3840), (1080, 1920))[int(rng.integers(0, 2))]
    g = torch.Generator(device="cpu").manual_seed(int(rng.integers(0, 1 << 30)))
    t = torch.randint(0, 256, (2, h, w, 3), dtype=torch.uint8, generator=g).to(device)
    if seed % 2: t[:, : h // 3] = int(rng.integers(0, 4))                     # a dark band: small pixel values
    s = float(rng.uniform(0.7, 1.8))
    fused = T._scale_t(t, s)
    monkeypatch.setenv("IMGXF_RESAMPLE_NO_MFMA", "1")
    assert torch.equal(fused, T._scale_t(t, s)), ("scale", h, w, s)
    monkeypatch.delenv("IMGXF_RESAMPLE_NO_MFMA")
    radius = float(rng.choice([2.0, 2.5, 3.0, 3.5, 4.0, 4.5, 5.0]))
    k = O.blur_ksize(radius)
    fx = ops.gaussian_blur(t, k, radius, fixed_point=True)
    fl = ops.gaussian_blur(t, k, radius)
    monkeypatch.setenv("IMGXF_FX_MFMA_MIN_R", "99"); monkeypatch.setenv("IMGXF_MFMA_MIN_R", "99")
    assert torch.equal(fx, ops.gaussian_blur(t, k, radius, fixed_point=True)), ("fixed", k)
    d = (fl.int() - ops.gaussian_blur(t, k, radius).int()).abs()
    assert int(d.max()) <= 1 and float((d != 0).float().mean()) < 1e-3, ("float", k)
    monkeypatch.delenv("IMGXF_FX_MFMA_MIN_R"); monkeypatch.delenv("IMGXF_MFMA_MIN_R")
    tx, ty = int(rng.integers(-w, w)), int(rng.integers(-h, h))
    ref = ops.new(t, h, w, (0, 0, 0))
    cl, ct, cr, cb = max(0, -tx), max(0, -ty), min(w, w - tx), min(h, h - ty)
    if cl < cr and ct < cb: ops.copy_rect(t, ref, cl, ct, max(0, tx), max(0, ty), cr - cl, cb - ct)
    assert torch.equal(T._translation_t(t, tx, ty), ref), ("translation", tx, ty)


def _jpeg_case(rng):
    """A random frame batch for the JPEG writer: sizes around every block / MCU boundary, contents from flat to the
    patterns that drive the DCT and the run / size symbols to their extremes."""
    h = int(rng.choice([1, 2, 7, 8, 9, 15, 16, 17, 31, 32, 33, int(rng.integers(1, 120))]))
    w = int(rng.choice([1, 2, 7, 8, 9, 15, 16, 17, 31, 255, 256, 257, 272, int(rng.integers(1, 400))]))
    n = int(rng.integers(1, 4))
    kind = int(rng.integers(0, 8))
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == 0:
        a = rng.integers(0, 256, (n, h, w, 3))
    elif kind == 1:                                   # 1-px checkerboard of extremes: the largest AC magnitudes
        a = np.broadcast_to((((yy + xx) & 1) * 255)[None, :, :, None], (n, h, w, 3))
    elif kind == 2:                                   # binary noise
        a = rng.integers(0, 2, (n, h, w, 3)) * 255
    elif kind == 3:                                   # smooth gradients: long zero runs, EOB early
        a = np.broadcast_to(((xx * int(rng.integers(1, 5)) + yy * int(rng.integers(0, 5))) % 256)[None, :, :, None], (n, h, w, 3))
    elif kind == 4:                                   # sparse impulses: ZRL chains
        a = np.full((n, h, w, 3), int(rng.integers(0, 256)))
        m = rng.random((n, h, w, 1)) < 0.01
        a = np.where(m, 255 - a, a)
    elif kind == 5:                                   # stripes of extremes along one axis
        a = np.broadcast_to((((xx // int(rng.integers(1, 4))) & 1) * 255)[None, :, :, None], (n, h, w, 3))
    elif kind == 6:                                   # per-channel flats with one saturated channel
        a = np.zeros((n, h, w, 3), np.int64)
        a[..., int(rng.integers(0, 3))] = 255
    else:                                             # low-amplitude noise around a level
        a = int(rng.integers(0, 250)) + rng.integers(0, 6, (n, h, w, 3))
    return np.ascontiguousarray(a, dtype=np.uint8), int(rng.choice([1, 5, 25, 50, 75, 75, 75, 90, 100]))


@pytest.mark.parametrize("seed", range(max(2, SEEDS // 4)))
def test_soak_jpeg_writer_equals_pillow(device, seed):
    import io
    from imagetransformations_amd import jpeg
    rng = np.random.default_rng(770000 + seed)
    a, quality = _jpeg_case(rng)
    t = torch.from_numpy(a).to(device)
    if seed % 3 == 1:                                  # a strided, unaligned view of a larger allocation
        big = torch.zeros((a.shape[0], a.shape[1] + 3, a.shape[2] + 5, 3), dtype=torch.uint8, device=device)
        big[:, 2:2 + a.shape[1], 1:1 + a.shape[2]] = t
        t = big[:, 2:2 + a.shape[1], 1:1 + a.shape[2]]
    files = jpeg.encode(t, quality)
    for i in range(a.shape[0]):
        buf = io.BytesIO()
        Image.fromarray(a[i]).save(buf, "JPEG", quality=quality)
        assert files[i] == buf.getvalue(), (seed, i, a.shape, quality)


@pytest.mark.parametrize("seed", range(SEEDS))
def test_soak_jpeg_reader_equals_pillow(device, seed):
    """Round 3: the device JPEG reader on random files — sizes 1 … 300, gray / 4:4:4 / 4:2:2 / 4:2:0, qualities 1 … 100,
    standard or optimised tables, with or without restart intervals, several files (equal and different sizes) per batch —
    against Pillow's decoder, bit for bit."""
    import io
    from imagetransformations_amd import jpeg_decode
    rng = np.random.default_rng(990000 + seed)
    files = []
    for _ in range(int(rng.integers(1, 6))):
        h, w = int(rng.integers(1, 300)), int(rng.integers(1, 300))
        if files and rng.random() < 0.3:
            h, w = prev
        prev = (h, w)
        a = _img(rng, h, w)
        kw = dict(quality=int(rng.integers(1, 101)))
        gray = rng.random() < 0.2
        if not gray:
            kw["subsampling"] = int(rng.integers(0, 3))
        if rng.random() < 0.3:
            kw["optimize"] = True
        r = rng.random()
        if r < 0.2:
            kw["restart_marker_blocks"] = int(rng.integers(1, 9))
        elif r < 0.4:
            kw["restart_marker_rows"] = int(rng.integers(1, 4))
        buf = io.BytesIO()
        try:
            (Image.fromarray(a).convert("L") if gray else Image.fromarray(a)).save(buf, "JPEG", **kw)
        except OSError:          # Pillow's ENCODER gives up on some optimise + restart combinations ("Suspension not allowed here")
            continue
        files.append(buf.getvalue())
    frames = jpeg_decode.decode(files, device)
    for k, (t, f) in enumerate(zip(frames, files)):
        want = np.asarray(Image.open(io.BytesIO(f)).convert("RGB"))
        assert np.array_equal(t.cpu().numpy(), want), (seed, k, want.shape)
