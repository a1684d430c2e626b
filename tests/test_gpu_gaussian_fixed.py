"""GPU: the opt-in fixed-point Gaussian (OpenCV's uint8 evaluation order, restated — parity
unpinned, see oracle.gaussian_blur_cv_fixed) through the C-ABI against the oracle, bit for bit,
on every kernel family behind it (marching, 4-byte marching, LDS-tiled)."""
import numpy as np
import pytest
import torch

from conftest import synth
from oracle import imgxf_oracle as O

pytestmark = pytest.mark.gpu
Image = pytest.importorskip("PIL.Image")

SHAPES = [
    (96, 1024, 3),     # 16-byte aligned rows > 1 KiB: marching kernel (k <= 9), 4-byte marching (k >= 11)
    (70, 352, 3),      # aligned, short rows: 4-byte marching
    (37, 61, 3),       # unaligned: LDS-tiled kernel
    (64, 2048, 1),
    (48, 512, 4),
    (33, 35, 1),
    (5, 7, 3),
]


@pytest.mark.parametrize("shape", SHAPES)
def test_gaussian_cv_fixed_matches_oracle(device, shape):
    from imagetransformations_amd import ops
    h, w, c = shape
    a = synth(h * 7 + w, h, w, c)
    t = torch.from_numpy(a).to(device)
    for ksize, sigma in ((3, 0.5), (5, 5 / 6), (7, 1.0), (9, 1.5), (13, 2.0), (19, 3.0), (31, 5.0), (5, 0.0), (7, -1.0), (9, 0.0), (1, 0.3)):
        if ksize // 2 + 1 > min(h, w):
            continue
        got = ops.gaussian_blur(t, ksize, sigma, fixed_point=True).cpu().numpy()
        assert np.array_equal(got, O.gaussian_blur_cv_fixed(a, ksize, sigma)), (shape, ksize, sigma)


def test_fixed_taps_entry_point(device):
    """imgxf_sepconv_fixed_u8 with the oracle's integer kernel == the Gaussian entry point; an
    asymmetric kernel goes through the tile kernel; sums above 256 are refused."""
    from imagetransformations_amd import ops
    a = synth(3, 80, 1024)
    t = torch.from_numpy(a).to(device)
    for ksize, sigma in ((5, 5 / 6), (11, 2.0)):
        k = [int(v) for v in O.gaussian_kernel_cv_fixed(ksize, sigma)]
        assert sum(k) == 256
        assert torch.equal(ops.sepconv_fixed(t, k, k), ops.gaussian_blur(t, ksize, sigma, fixed_point=True))
    kx, ky = [10, 200, 46], [0, 0, 128, 100, 28]
    got = ops.sepconv_fixed(t, kx, ky).cpu().numpy()
    p = np.pad(a.astype(np.int64), ((2, 2), (1, 1), (0, 0)), mode="reflect")
    hp = sum(kx[j] * p[:, j:j + a.shape[1]] for j in range(3))
    vp = sum(ky[i] * hp[i:i + a.shape[0]] for i in range(5))
    assert np.array_equal(got, ((vp + 32768) >> 16).astype(np.uint8))
    with pytest.raises(Exception):
        ops.sepconv_fixed(t, [100, 100, 100], [0, 256, 0])
    with pytest.raises(ValueError):
        ops.gaussian_blur(t, 5, 1.0, return_f32=True, fixed_point=True)


def test_fixed_differs_from_float_by_small_steps_only(device):
    from imagetransformations_amd import ops
    a = synth(11, 128, 1024)
    t = torch.from_numpy(a).to(device)
    d = (ops.gaussian_blur(t, 5, 5 / 6, fixed_point=True).int() - ops.gaussian_blur(t, 5, 5 / 6).int()).abs()
    assert int(d.max()) <= 2 and 0 < float((d > 0).float().mean()) < 0.2


def test_facade_switch(device):
    from imagetransformations_amd import transformation as T
    a = synth(21, 64, 96)
    img = Image.fromarray(a)
    try:
        T.BLUR_FIXED_POINT = True
        got = np.asarray(T.apply_blur(img, 1.5))
    finally:
        T.BLUR_FIXED_POINT = False
    assert np.array_equal(got, O.gaussian_blur_cv_fixed(a, O.blur_ksize(1.5), 1.5))
    assert np.array_equal(np.asarray(T.apply_blur(img, 1.5)), O.apply_blur(a, 1.5))


def test_float_path_with_sigma_not_given(device):
    """sigma <= 0: OpenCV's binomial tables for ksize <= 7, its sigma formula beyond."""
    from imagetransformations_amd import ops
    for shape in ((64, 1024, 3), (37, 61, 3)):
        a = synth(5, *shape)
        t = torch.from_numpy(a).to(device)
        for ksize in (3, 5, 7, 9, 15):
            d = np.abs(ops.gaussian_blur(t, ksize, 0.0).cpu().numpy().astype(int) - O.gaussian_blur(a, ksize, 0.0).astype(int))
            assert d.max() <= 1 and (d != 0).mean() < 1e-3, (shape, ksize)
        # the 1-2-1 kernel has exact ties (x.5): both sides round half to even
        assert np.array_equal(ops.gaussian_blur(t, 3, -1.0).cpu().numpy(), O.gaussian_blur(a, 3, -1.0))


@pytest.mark.parametrize("w", [96, 112, 176, 336, 352, 1280, 2000])
def test_large_kernels_across_strip_geometries(device, w):
    """k >= 11 on RGB takes the pixel-stride marching kernel (252-byte strips, 21 pixel groups per
    wave): widths around the strip boundaries, several frames, fixed-point bit for bit and the float
    path to 1e-5 relative before rounding."""
    from imagetransformations_amd import ops
    h = 70
    frames = np.stack([synth(w + i, h, w) for i in range(3)])
    t = torch.from_numpy(frames).to(device)
    for ksize, sigma in ((11, 11 / 6), (13, 2.0), (17, 2.9), (21, 3.5), (25, 4.0), (31, 5.0)):
        if w * 3 < 256 + 32 * ((ksize // 2 * 3 + 15) // 16):
            continue
        got = ops.gaussian_blur(t, ksize, sigma, fixed_point=True).cpu().numpy()
        out, f32 = ops.gaussian_blur(t, ksize, sigma, return_f32=True)
        for i in range(3):
            assert np.array_equal(got[i], O.gaussian_blur_cv_fixed(frames[i], ksize, sigma)), (w, ksize, i)
            ref = O.gaussian_blur_f64(frames[i], ksize, sigma)
            assert (np.abs(f32[i].cpu().numpy() - ref) <= 1e-5 * np.maximum(np.abs(ref), 1.0)).all(), (w, ksize, i)


@pytest.mark.parametrize("hw", [(32, 96), (37, 352), (270, 480), (65, 1280), (129, 112), (200, 2048)])
def test_fixed_point_on_the_i8_matrix_cores(device, monkeypatch, hw):
    """k >= 11 on aligned RGB rows: sepconv_fx_mfma.inc (integer band products), bit-exact with the oracle and
    with the vector kernels; row chunks, batches, asymmetric integer taps."""
    from imagetransformations_amd import ops
    h, w = hw
    a = np.stack([synth(600 + i, h, w) for i in range(2)])
    t = torch.from_numpy(a).to(device)
    for ksize, sigma in ((11, 2.0), (13, 2.0), (15, 2.5), (19, 3.0), (21, 3.5), (25, 4.0), (27, 4.5), (31, 5.0)):
        got = ops.gaussian_blur(t, ksize, sigma, fixed_point=True)
        for i in range(2):
            assert np.array_equal(got[i].cpu().numpy(), O.gaussian_blur_cv_fixed(a[i], ksize, sigma)), (hw, ksize, i)
        monkeypatch.setenv("IMGXF_FX_MFMA_MIN_R", "99")
        assert torch.equal(got, ops.gaussian_blur(t, ksize, sigma, fixed_point=True)), (hw, ksize)
        monkeypatch.delenv("IMGXF_FX_MFMA_MIN_R")
    monkeypatch.setenv("IMGXF_FX_MFMA_MIN_R", "2")          # small radii through the same kernel
    for ksize, sigma in ((5, 2.0), (7, 2.5), (9, 3.0)):
        assert np.array_equal(ops.gaussian_blur(t, ksize, sigma, fixed_point=True)[0].cpu().numpy(),
                              O.gaussian_blur_cv_fixed(a[0], ksize, sigma)), (hw, ksize)
    for bpc in ("1", "3"):
        monkeypatch.setenv("IMGXF_MFMA2_BPC", bpc)
        assert np.array_equal(ops.gaussian_blur(t, 19, 3.0, fixed_point=True)[1].cpu().numpy(), O.gaussian_blur_cv_fixed(a[1], 19, 3.0)), bpc
    monkeypatch.delenv("IMGXF_MFMA2_BPC")
    monkeypatch.delenv("IMGXF_FX_MFMA_MIN_R")
    kx = [3, 9, 20, 40, 60, 50, 35, 20, 10, 6, 3]          # asymmetric, sums to 256; ky symmetric
    ky = [1, 4, 10, 22, 40, 102, 40, 22, 10, 4, 1]
    got = ops.sepconv_fixed(t[0], kx, ky).cpu().numpy()
    p = np.pad(a[0].astype(np.int64), ((5, 5), (5, 5), (0, 0)), mode="reflect")
    hp = sum(kx[j] * p[:, j:j + w] for j in range(11))
    vp = sum(ky[i] * hp[i:i + h] for i in range(11))
    assert np.array_equal(got, ((vp + 32768) >> 16).astype(np.uint8))
