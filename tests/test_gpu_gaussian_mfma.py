"""Large-radius Gaussians on the matrix cores (csrc/sepconv_mfma.inc): both separable passes as
banded-Toeplitz products on v_mfma_f32_32x32x16_f16 (bytes as exact f16 values, weights split in two
f16 halves, fp32 accumulation).  The pre-quantisation value must stay within the 1e-5 relative
contract of the float definition of cv2.GaussianBlur (/root/reference/transformation.py:249) for
every kernel size the reference's blur grid produces (k = 13 ... 31), at image borders, for short
images (one or two 32-row blocks), partial column tiles, batches and strided views."""
import numpy as np
import pytest
import torch

from conftest import synth
from oracle import imgxf_oracle as O
from test_gpu_parity import MFMA_ABS, assert_quantised_close, dev, host

pytestmark = pytest.mark.gpu

RADII = [2.0, 2.5, 3.0, 3.5, 4.0, 4.5, 5.0]          # transformation.py:102 -> k = 13, 15, 19, 21, 25, 27, 31


@pytest.mark.parametrize("hw", [(32, 96), (37, 352), (270, 480), (65, 1280), (129, 112)])
@pytest.mark.parametrize("radius", RADII)
def test_mfma_gaussian_within_contract(device, hw, radius):
    from imagetransformations_amd import ops
    a = synth(31, *hw)
    k = O.blur_ksize(radius)
    out, f32 = ops.gaussian_blur(dev(a, device), k, radius, return_f32=True)
    assert_quantised_close(host(out), host(f32), O.gaussian_blur_f64(a, k, radius), O.saturate_u8, abs_term=MFMA_ABS * float(a.max()))


@pytest.mark.parametrize("radius", [1.0, 1.5, 2.0, 5.0])
def test_mfma_small_radii_batches_and_views(device, monkeypatch, radius):
    from imagetransformations_amd import ops
    monkeypatch.setenv("IMGXF_MFMA_MIN_R", "2")
    n, h, w = 11, 70, 480
    a = np.stack([synth(40 + i, h, w) for i in range(n)])
    k = O.blur_ksize(radius)
    out, f32 = ops.gaussian_blur(dev(a, device), k, radius, return_f32=True)
    out, f32 = host(out), host(f32)
    for i in range(n):
        assert_quantised_close(out[i], f32[i], O.gaussian_blur_f64(a[i], k, radius), O.saturate_u8, abs_term=MFMA_ABS * float(a[i].max()))
    big = dev(np.stack([synth(60 + i, h, w + 16) for i in range(6)]), device)
    view = big[::2, :, :w]                                   # strided frames, padded rows
    got = host(ops.gaussian_blur(view, k, radius))
    assert np.array_equal(got, host(ops.gaussian_blur(view.contiguous(), k, radius)))
    # constant images are fixed points (the weights' two halves sum to 1 within 2^-22)
    const = torch.full((2, 64, 256, 3), 201, dtype=torch.uint8, device=device)
    assert bool((ops.gaussian_blur(const, k, radius) == 201).all())


def test_mfma_equals_vector_kernel_up_to_rounding(device, monkeypatch):
    """Same op through the vector-pipe kernel (IMGXF_MFMA_MIN_R beyond every radius): at most 1 LSB apart on
    rounding ties."""
    from imagetransformations_amd import ops
    a = dev(np.stack([synth(80 + i, 200, 640) for i in range(3)]), device)
    for radius in (2.0, 5.0):
        k = O.blur_ksize(radius)
        m = host(ops.gaussian_blur(a, k, radius))
        monkeypatch.setenv("IMGXF_MFMA_MIN_R", "99")
        v = host(ops.gaussian_blur(a, k, radius))
        monkeypatch.delenv("IMGXF_MFMA_MIN_R")
        d = np.abs(m.astype(int) - v.astype(int))
        assert d.max() <= 1 and (d != 0).mean() < 1e-3


@pytest.mark.parametrize("hw", [(37, 352), (129, 112), (270, 480)])
@pytest.mark.parametrize("radius", [2.0, 3.5, 5.0])
def test_lds_staged_mfma_kernel_still_within_contract(device, monkeypatch, hw, radius):
    """IMGXF_MFMA_V1 selects the first structure (LDS-staged 128-byte tiles); the default is the wave-owned one."""
    from imagetransformations_amd import ops
    monkeypatch.setenv("IMGXF_MFMA_V1", "1")
    a = synth(33, *hw)
    k = O.blur_ksize(radius)
    out, f32 = ops.gaussian_blur(dev(a, device), k, radius, return_f32=True)
    assert_quantised_close(host(out), host(f32), O.gaussian_blur_f64(a, k, radius), O.saturate_u8, abs_term=MFMA_ABS * float(a.max()))


def test_mfma_row_chunks_and_full_size_agree(device, monkeypatch):
    """Chunk length only changes which workgroup computes a row; 4K frames against the vector kernel."""
    from imagetransformations_amd import ops
    a = dev(np.stack([synth(90 + i, 300, 512) for i in range(2)]), device)
    k = O.blur_ksize(4.0)
    ref = host(ops.gaussian_blur(a, k, 4.0))
    for bpc in ("1", "2", "3", "100"):
        monkeypatch.setenv("IMGXF_MFMA2_BPC", bpc)
        assert np.array_equal(host(ops.gaussian_blur(a, k, 4.0)), ref), bpc
    monkeypatch.delenv("IMGXF_MFMA2_BPC")
    g = torch.Generator(device="cpu").manual_seed(3)
    t = torch.randint(0, 256, (2, 2160, 3840, 3), dtype=torch.uint8, generator=g).to(device)
    for radius in (2.0, 5.0):
        k = O.blur_ksize(radius)
        m = ops.gaussian_blur(t, k, radius)
        monkeypatch.setenv("IMGXF_MFMA_MIN_R", "99")
        v = ops.gaussian_blur(t, k, radius)
        monkeypatch.delenv("IMGXF_MFMA_MIN_R")
        d = (m.int() - v.int()).abs()
        assert int(d.max()) <= 1 and float((d != 0).float().mean()) < 1e-3


def test_mfma_gaussian_random_geometries(device):
    """Seeded sweep over sizes (16-byte rows), radii, batch sizes and anisotropic sigmas against the fp64 oracle."""
    from imagetransformations_amd import ops
    rng = np.random.default_rng(77)
    for it in range(24):
        h, w = int(rng.integers(32, 200)), int(rng.integers(6, 40)) * 16
        radius = float(rng.choice(RADII))
        k = O.blur_ksize(radius)
        n = int(rng.integers(1, 4))
        a = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
        if it % 4 == 0: a[:, :, : w // 3] = 255
        out, f32 = ops.gaussian_blur(torch.from_numpy(a).to(device), k, radius, return_f32=True)
        out, f32 = host(out), host(f32)
        for i in range(n):
            assert_quantised_close(out[i], f32[i], O.gaussian_blur_f64(a[i], k, radius), O.saturate_u8, abs_term=MFMA_ABS * float(a[i].max()))
