"""cv2.filter2D with rank-1 kernels (TransformationPool.motion_blur,
/root/reference/pipenline/cifar_image_transformations.py:109-119; BASELINE configs[0]'s 3x3 box) runs on
the separable kernels: the factorised filter must satisfy the same fp64-oracle contract as the dense
kernel, for asymmetric taps and every kernel family the dispatcher can pick (marching, 4-byte marching,
matrix cores, tiled), and agree with the dense evaluation up to rounding ties."""
import numpy as np
import pytest
import torch

from conftest import synth
from oracle import imgxf_oracle as O
from test_gpu_parity import dev, host

pytestmark = pytest.mark.gpu


def _check(out, ref_f):
    ref = O.saturate_u8(ref_f)
    diff = np.abs(out.astype(int) - ref.astype(int))
    near_tie = np.abs(ref_f - np.floor(ref_f) - 0.5) < 1e-4
    assert diff.max() <= 1 and (diff == 0)[~near_tie].all()


@pytest.mark.parametrize("hw", [(32, 32), (37, 61), (64, 352), (270, 480), (129, 1280), (5, 7), (3, 3), (2, 9), (1, 1)])   # the last ones: images smaller than the kernel
@pytest.mark.parametrize("size", [5, 7, 9, 11, 13, 15])
def test_motion_blur_rows_every_family(device, hw, size):
    from imagetransformations_amd import ops
    a = synth(140 + size, *hw)
    out = host(ops.conv2d(dev(a, device), O.motion_blur_kernel(size)))
    _check(out, O.conv2d_f64(a, O.motion_blur_kernel(size)))


@pytest.mark.parametrize("hw", [(48, 64), (200, 352), (66, 1280)])
def test_asymmetric_outer_products_and_generic_sepconv(device, hw):
    from imagetransformations_amd import ops
    rng = np.random.default_rng(5)
    a = synth(150, *hw)
    t = dev(a, device)
    for nkx, nky in [(3, 5), (7, 3), (9, 9), (11, 5), (13, 15), (15, 1), (1, 7)]:
        kx = rng.uniform(-0.2, 0.5, nkx).astype(np.float32)
        ky = rng.uniform(-0.2, 0.5, nky).astype(np.float32)
        kx /= np.float32(abs(kx.sum()) + 0.5); ky /= np.float32(abs(ky.sum()) + 0.5)
        k2 = np.outer(ky.astype(np.float64), kx.astype(np.float64))
        ref_f = O.conv2d_f64(a, k2)
        _check(host(ops.sepconv(t, kx.tolist(), ky.tolist())), ref_f)          # the separable entry itself
        _check(host(ops.conv2d(t, k2.tolist())), ref_f)                        # filter2D recognising the outer product


def test_non_separable_kernels_keep_the_dense_path(device, monkeypatch):
    from imagetransformations_amd import ops
    a = synth(160, 96, 160)
    t = dev(a, device)
    diag = (np.eye(5) / 5.0)
    _check(host(ops.conv2d(t, diag.tolist())), O.conv2d_f64(a, diag))          # rank 5, sparse: zero taps are skipped
    lap = np.array([[0, -1, 0], [-1, 5, -1], [0, -1, 0]], np.float64)
    _check(host(ops.conv2d(t, lap.tolist())), O.conv2d_f64(a, lap))
    # separable and dense evaluation of the same rank-1 kernel: rounding ties only
    box = O.box_kernel(5)
    s = host(ops.conv2d(t, box.tolist()))
    monkeypatch.setenv("IMGXF_CONV2D_NO_SEPARABLE", "1")
    d = host(ops.conv2d(t, box.tolist()))
    dd = np.abs(s.astype(int) - d.astype(int))
    assert dd.max() <= 1 and (dd != 0).mean() < 1e-3


def test_motion_blur_full_size_batch(device, monkeypatch):
    from imagetransformations_amd import ops
    g = torch.Generator(device="cpu").manual_seed(11)
    t = torch.randint(0, 256, (3, 1080, 1920, 3), dtype=torch.uint8, generator=g).to(device)
    for size in (5, 11):
        k = O.motion_blur_kernel(size).tolist()
        s = ops.conv2d(t, k)
        monkeypatch.setenv("IMGXF_CONV2D_NO_SEPARABLE", "1")
        d = ops.conv2d(t, k)
        monkeypatch.delenv("IMGXF_CONV2D_NO_SEPARABLE")
        dd = (s.int() - d.int()).abs()
        assert int(dd.max()) <= 1 and float((dd != 0).float().mean()) < 1e-3


@pytest.mark.parametrize("k", [13, 17, 21, 31])
def test_unnormalised_taps_stay_off_the_f16_matrix_cores(device, k):
    """the f16 matrix-core Gaussian only holds for taps below 2 and row sums below 256
    (mfma_eligible checks the taps); 2 * ones(k), integer binomial rows and huge row sums must take
    the vector kernels and satisfy the same fp64 contract (results saturate at 255 / clip at 0)."""
    from imagetransformations_amd import ops
    a = synth(170 + k, 96, 352)
    a[:20] //= 64                                   # small values so that the big taps do not saturate everything
    t = dev(a, device)
    cases = [(np.full(k, 2.0), np.full(k, 1.0 / (2 * k * k))),                 # |w| >= 2 on x
             (np.full(k, 1.0 / (2 * k * k)), np.full(k, 2.5)),                 # ... on y
             (np.full(k, 30.0), np.full(k, 1.0 / (30.0 * k * k))),             # sum|wx| > 256
             (np.r_[-3.0, np.zeros(k - 2), 3.0], np.r_[np.zeros(k // 2), 0.25, np.zeros(k // 2)])]
    for kx, ky in cases:
        kx32, ky32 = kx.astype(np.float32), ky.astype(np.float32)
        ref_f = O.conv2d_f64(a, np.outer(ky32.astype(np.float64), kx32.astype(np.float64)))
        out, f32 = ops.sepconv(t, kx32.tolist(), ky32.tolist(), return_f32=True)
        got = f32.cpu().numpy()
        assert np.isfinite(got).all()
        assert (np.abs(got - ref_f) <= 1e-5 * np.maximum(np.abs(ref_f), 1.0)).all()
        _check(host(out), ref_f)
    if k <= 15:                                                               # imgxf_conv2d_u8 takes kernels up to 15 x 15
        box2 = 2.0 * np.ones((k, k)) / (k * k)                               # filter2D form of a rank-1 kernel
        _check(host(ops.conv2d(t, box2.tolist())), O.conv2d_f64(a, box2))
