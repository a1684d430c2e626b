"""Seeded random geometries through the fast paths added late in round 1 (marching dilation,
RGB composite, LDS-staged Lanczos, fp32 shear with fp64 hand-back, LDS-DMA nearest rotation,
marching Gaussian with static slots, strided batch views) against the oracle."""
import numpy as np
import pytest
import torch

from oracle import imgxf_oracle as O

pytestmark = pytest.mark.gpu


def dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def host(t):
    return t.cpu().numpy()


def rnd_image(rng, h, w, c=3):
    kind = rng.integers(0, 3)
    if kind == 0:
        return rng.integers(0, 256, (h, w, c), dtype=np.uint8)
    if kind == 1:                                    # smooth ramps + noise: many near-integer interpolants
        y, x = np.mgrid[0:h, 0:w]
        base = (x * 255 // max(w - 1, 1) + y * 3) % 256
        return np.stack([(base + k * 40 + rng.integers(0, 3, (h, w))) % 256 for k in range(c)], -1).astype(np.uint8)
    a = np.zeros((h, w, c), np.uint8)                # flat regions with sparse spikes
    a[rng.random((h, w)) < 0.02] = rng.integers(1, 256, c)
    return a


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_aligned_fast_paths(device, seed):
    from imagetransformations_amd import ops
    rng = np.random.default_rng(1000 + seed)
    h = int(rng.integers(40, 200))
    w = int(rng.integers(8, 90)) * 16                # 16-pixel multiples: rows are 16-byte aligned for c = 1 and 3
    a = rnd_image(rng, h, w)
    t = dev(a, device)
    # nearest rotation (LDS-DMA staging) and bilinear precise
    ang = float(rng.uniform(-180, 180))
    assert np.array_equal(host(ops.rotate(t, -ang, ops.NEAREST, (0, 0, 0))), O.apply_rotation(a, ang)), ("nearest", h, w, ang)
    m = O.rotate_zoom_matrix(w, h, float(rng.uniform(-60, 60)), float(rng.uniform(0.7, 2.0)))
    assert np.array_equal(host(ops.affine(t, m, (w, h), ops.BILINEAR, (1, 2, 3), precise=True)),
                          O.affine_bilinear(a, (w, h), m, fill=(1, 2, 3))), ("bilinear", h, w, m)
    # shear (fp32 cubic + fp64 hand-back)
    sh = float(rng.choice([0.1, 0.2, 0.30000000000000004, 0.5, 0.8, 1.0]))
    nw, ms = O.shear_geometry(w, h, sh)
    assert np.array_equal(host(ops.affine(t, ms, (nw, h), ops.BICUBIC, (255, 255, 255), precise=True)), O.apply_shear(a, sh)), ("shear", h, w, sh)
    # Lanczos / bicubic resize through the LDS-staged horizontal pass
    sc = float(rng.uniform(0.6, 1.6))
    size = (max(1, int(w * sc)), max(1, int(h * sc)))
    for flt in (O.RESAMPLE_LANCZOS, O.RESAMPLE_BICUBIC):
        assert np.array_equal(host(ops.resize(t, size, flt)), O.resize(a, size, flt)), ("resize", h, w, size, flt)
    # Gaussian (marching kernels where the row is wider than 1 KiB, tiled otherwise)
    r = float(rng.choice([0.5, 5 / 6, 1.0, 1.5, 2.0, 3.5]))
    k = O.blur_ksize(r)
    if h > k and w > k:
        out, f32 = ops.gaussian_blur(t, k, r, return_f32=True)
        ref = O.gaussian_blur_f64(a, k, r)
        mfma = 0.0      # (the matrix-core kernels hold the plain tolerance since their bytes go in as b * 2^-22: DESIGN section 4)
        assert (np.abs(host(f32) - ref) <= 1e-5 * np.maximum(np.abs(ref), 1.0) + mfma).all(), ("gauss", h, w, k)
    # mask stage
    g = a[..., 0]
    mask = g > np.percentile(g, 90)
    mt = dev((mask * rng.integers(1, 256)).astype(np.uint8), device)
    assert np.array_equal(host(ops.dilate_cross(mt, 3)), (O.binary_dilation_cross(mask, 3) * 255).astype(np.uint8)), ("dilate", h, w)
    b = rnd_image(rng, h, w)
    fg = (O.binary_dilation_cross(mask, 3) * 255).astype(np.uint8)
    assert np.array_equal(host(ops.composite(t, dev(b, device), dev(fg, device))), O.composite(a, b, fg)), ("composite", h, w)
    bg = host(ops.new(t, h, w, (7, 130, 251)))
    assert (bg == np.array([7, 130, 251], np.uint8)).all()


@pytest.mark.parametrize("seed", range(3))
def test_fuzz_strided_batches(device, seed):
    """Every other frame of a batch, and a column window of wider frames (row stride > row bytes)."""
    from imagetransformations_amd import ops
    rng = np.random.default_rng(2000 + seed)
    h, w = int(rng.integers(30, 80)), int(rng.integers(5, 20)) * 16
    big = np.stack([rnd_image(rng, h, w + 32) for _ in range(4)])
    tb = dev(big, device)
    win = tb[::2, :, 16:16 + w]                                   # frame stride x2, row stride w+32, 48-byte column offset
    ref = [np.ascontiguousarray(big[i, :, 16:16 + w]) for i in (0, 2)]
    ang = float(rng.uniform(-90, 90))
    out = host(ops.rotate(win, -ang, ops.NEAREST, (0, 0, 0)))
    lz = host(ops.resize_lanczos(win, (w + 9, h - 5)))
    g = host(ops.gaussian_blur(win, 5, 5 / 6))
    sm = host(ops.solarize(win, 99))
    eq = host(ops.equalize(win))
    for j in range(2):
        assert np.array_equal(out[j], O.apply_rotation(ref[j], ang))
        assert np.array_equal(lz[j], O.resize_lanczos(ref[j], (w + 9, h - 5)))
        d = np.abs(g[j].astype(int) - O.gaussian_blur(ref[j], 5, 5 / 6).astype(int))
        assert d.max() <= 1 and (d != 0).mean() < 1e-3
        assert np.array_equal(sm[j], O.solarize(ref[j], 99))
        assert np.array_equal(eq[j], O.equalize(ref[j]))


@pytest.mark.parametrize("hw", [(130, 256), (64, 64), (300, 480)])
def test_bilinear_special_matrices(device, hw):
    """Matrices that stress the interior / border-list split of the bilinear kernels: pure scale,
    pure translation (partly and completely outside), anisotropic zoom-out, shear-like, a
    different output size, and a flip (negative diagonal)."""
    from imagetransformations_amd import ops
    h, w = hw
    rng = np.random.default_rng(h * 31 + w)
    a = rnd_image(rng, h, w)
    t = dev(a, device)
    mats = [
        ((0.5, 0, 0, 0, 0.5, 0), (w, h)),                       # 2x zoom-in of the top-left quarter
        ((1, 0, 10.25, 0, 1, -7.5), (w, h)),                    # translation with fractions
        ((1, 0, 5 * w, 0, 1, 0), (w, h)),                       # completely outside: all fill
        ((1.7, 0, -20, 0, 2.3, -30), (w, h)),                   # anisotropic zoom-out with borders
        ((1, 0.3, -15, 0.1, 1, 0), (w + 40, h)),                # shear-like, wider output
        ((-1, 0, w, 0, 1, 0), (w, h)),                          # mirror
        ((0.9, 0.05, 3.3, -0.04, 1.1, 2.2), (w // 2 + 3, h // 2 + 5)),   # small output
    ]
    for m, size in mats:
        for fill in ((0, 0, 0), (9, 200, 77)):
            want = O.affine_bilinear(a, size, m, fill=fill)
            got = host(ops.affine(t, m, size, ops.BILINEAR, fill, precise=True))
            assert np.array_equal(got, want), (hw, m, size, fill, int((got != want).sum()))
        # batch of two frames through the same launch
        batch = dev(np.stack([a, a[::-1].copy()]), device)
        got2 = host(ops.affine(batch, m, size, ops.BILINEAR, (1, 2, 3), precise=True))
        assert np.array_equal(got2[1], O.affine_bilinear(a[::-1].copy(), size, m, fill=(1, 2, 3))), (hw, m)


@pytest.mark.parametrize("seed", range(4))
def test_resize_crop_equals_resize_then_crop(device, seed):
    """ops.resize_crop (window fused into the resample plan) == Pillow's resize followed by crop,
    for every filter, aligned and unaligned widths, windows touching every border."""
    from imagetransformations_amd import ops
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3000 + seed)
    h = int(rng.integers(30, 120))
    w = int(rng.integers(2, 30)) * 16 if seed % 2 == 0 else int(rng.integers(30, 300))
    a = rnd_image(rng, h, w)
    t = dev(np.stack([a, a[::-1].copy()]), device)
    for sc in (1.1, 1.3, 0.8, 1.7):
        nw, nh = int(w * sc), int(h * sc)
        boxes = [((nw - w) // 2, (nh - h) // 2, (nw - w) // 2 + w, (nh - h) // 2 + h)] if sc > 1 else []
        boxes += [(0, 0, max(1, nw // 2), max(1, nh // 3)), (nw // 3, nh // 2, nw, nh), (nw - 1, nh - 1, nw, nh), (0, 0, nw, nh)]
        for box in boxes:
            for flt in (1, 3, 2):
                want0 = np.asarray(Image.fromarray(a).resize((nw, nh), flt).crop(box))
                want1 = np.asarray(Image.fromarray(a[::-1].copy()).resize((nw, nh), flt).crop(box))
                got = host(ops.resize_crop(t, (nw, nh), box, flt))
                assert np.array_equal(got[0], want0) and np.array_equal(got[1], want1), (h, w, sc, box, flt)


@pytest.mark.parametrize("hw", [(40, 64), (37, 61), (120, 1040)])
def test_fused_background_stages(device, hw):
    """rgb_sobel (L never stored) == sobel(rgb2l(.)) for all variants, composite_const ==
    composite with a materialised constant image."""
    from imagetransformations_amd import ops, _ffi
    h, w = hw
    rng = np.random.default_rng(h + w)
    a = rnd_image(rng, h, w)
    t = dev(np.stack([a, a[:, ::-1].copy()]), device)
    for variant in (_ffi.SOBEL_X_WRAP, _ffi.SOBEL_Y_WRAP, _ffi.SOBEL_MAGNITUDE):
        assert torch.equal(ops.rgb_sobel(t, variant), ops.sobel(ops.rgb2l(t), variant))
    assert np.array_equal(host(ops.rgb_sobel(t))[0, ..., 0], O.sobel_scipy(O.rgb2l(a)))
    mask = dev(((rng.random((2, h, w, 1)) < 0.3) * int(rng.integers(1, 256))).astype(np.uint8), device)
    want = ops.composite(t, ops.new(t, h, w, (10, 200, 30)), mask)
    assert torch.equal(ops.composite_const(t, (10, 200, 30), mask), want)


@pytest.mark.parametrize("seed", range(8))
def test_unit_step_bicubic_rows(device, seed):
    """Horizontal-only bicubic with m0 == 1 (the row-constant fast path of the shear kernel plus its
    row-end pass): random shear / shift incl. negative ones, integer row offsets, outputs narrower
    and wider than the source, tiny and 16-byte-aligned widths, fractions near 0 / 0.5 / 1."""
    from imagetransformations_amd import ops
    rng = np.random.default_rng(5000 + seed)
    h = int(rng.integers(3, 90))
    w = int(rng.choice([4, 5, 7, 9, 16, 33, 64, 257, 300, 1040]))
    a = rnd_image(rng, h, w)
    t = dev(a, device)
    for _ in range(6):
        a1 = float(rng.choice([0.0, 0.3, -0.3, 0.5, 1.0, 0.123456789, -0.77]))
        a2 = float(rng.choice([0.0, -3.0, 2.5, -0.5, 1e-12, 0.4999999999, float(rng.uniform(-w, w))]))
        m5 = float(rng.integers(-2, 3))
        ow = int(rng.choice([w, w + 13, max(1, w - 3), 2 * w + 1, 3]))
        m = (1.0, a1, a2, 0.0, 1.0, m5)
        want = O.affine_bicubic(a, (ow, h), m, fill=(255, 255, 255))
        got = host(ops.affine(t, m, (ow, h), ops.BICUBIC, (255, 255, 255), precise=True))
        assert np.array_equal(got, want), (h, w, m, ow, int((got != want).sum()))


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_perspective_and_fixed_point_blur(device, seed):
    """Random sizes / channel counts / coefficient draws through the perspective kernel (aligned and
    unaligned staging, partial tiles, per-frame coefficients) and the fixed-point Gaussian."""
    from imagetransformations_amd import ops
    rng = np.random.default_rng(5000 + seed)
    h, w = int(rng.integers(3, 260)), int(rng.integers(3, 400))
    c = int(rng.choice([1, 3, 4]))
    n = int(rng.integers(1, 4))
    frames = np.stack([rnd_image(rng, h, w, c) for _ in range(n)])
    if c == 1:
        frames = frames[..., 0]
    g = torch.Generator().manual_seed(seed)
    cs = []
    for i in range(n):
        ds = float(rng.choice([0.0, 0.1, 0.2, 0.35, 0.7]))
        st, en = O.perspective_endpoints(w, h, ds, lambda lo, hi: int(torch.randint(lo, hi, size=(1,), generator=g).item()))
        cs.append([float(v) for v in O.perspective_coeffs(st, en)])
    t = dev(frames, device) if c != 1 else dev(frames, device).unsqueeze(-1)
    got = host(ops.perspective(t, cs))
    for i in range(n):
        want = O.perspective_warp(frames[i], cs[i])
        assert np.array_equal(got[i] if c != 1 else got[i][..., 0], want), ("perspective", seed, h, w, c, i)
    r = float(rng.choice([0.5, 5 / 6, 1.0, 2.0, 3.5]))
    k = O.blur_ksize(r)
    if h > k and w > k:
        got = host(ops.gaussian_blur(t, k, r, fixed_point=True))
        for i in range(n):
            want = O.gaussian_blur_cv_fixed(frames[i], k, r)
            assert np.array_equal(got[i] if c != 1 else got[i][..., 0], want), ("fixed blur", seed, h, w, c, k)
