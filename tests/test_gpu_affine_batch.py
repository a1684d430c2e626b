"""Batched bilinear rotate / zoom (csrc/affine_mf.inc): the interior tiles of a batch loop over
the frames inside the workgroup with the per-pixel geometry held in registers.  Every frame of a
batch must equal the Pillow-exact oracle (Image.transform(AFFINE, BILINEAR); SURVEY §8a row
a2', benchmark configs[3]) bit for bit in precise mode, and the fp32 mode must stay within the
1e-5 contract; ragged frame groups, strided views and geometries that fall back to the
per-frame kernels are covered."""
import numpy as np
import pytest
import torch

from conftest import synth
from oracle import imgxf_oracle as O
from test_gpu_parity import dev, host

pytestmark = pytest.mark.gpu


def _matrices(w, h):
    return [O.rotate_zoom_matrix(w, h, 30.0, 1.5), O.rotate_zoom_matrix(w, h, -12.5, 1.1),
            O.rotate_zoom_matrix(w, h, 75.0, 2.0), O.rotate_plan(w, h, 7.0)[1],
            O.rotate_zoom_matrix(w, h, 200.0, 0.8)]          # the last one minifies: box too large for the tall tiles


@pytest.mark.parametrize("n,h,w", [(2, 96, 160), (5, 270, 480), (3, 334, 500), (9, 200, 320)])
def test_batched_bilinear_precise_bit_exact(device, n, h, w):
    from imagetransformations_amd import ops
    a = np.stack([synth(700 + i, h, w) for i in range(n)])
    t = dev(a, device)
    for m in _matrices(w, h):
        got = host(ops.affine(t, m, (w, h), ops.BILINEAR, (0, 0, 0), precise=True))
        for i in range(n):
            assert np.array_equal(got[i], O.affine_bilinear(a[i], (w, h), m, fill=(0, 0, 0))), (m, i)


def test_batched_bilinear_frames_per_block_and_views(device, monkeypatch):
    from imagetransformations_amd import ops
    n, h, w = 7, 270, 480
    a = np.stack([synth(720 + i, h, w) for i in range(n)])
    t = dev(a, device)
    m = O.rotate_zoom_matrix(w, h, 30.0, 1.5)
    want = np.stack([O.affine_bilinear(a[i], (w, h), m, fill=(9, 8, 7)) for i in range(n)])
    for fpb in ("1", "2", "3", "7", "16"):
        monkeypatch.setenv("IMGXF_AFFINE_FPB", fpb)
        assert np.array_equal(host(ops.affine(t, m, (w, h), ops.BILINEAR, (9, 8, 7), precise=True)), want), fpb
    monkeypatch.delenv("IMGXF_AFFINE_FPB")
    # round 3: 64 x 32 tiles are the default when their source box fits; the 32 x 64 tiles stay behind a knob (and
    # serve geometries whose wide box is too large)
    monkeypatch.setenv("IMGXF_AFFINE_MF_NARROW", "1")
    assert np.array_equal(host(ops.affine(t, m, (w, h), ops.BILINEAR, (9, 8, 7), precise=True)), want)
    for mm in _matrices(w, h):
        got = host(ops.affine(t[:3], mm, (w, h), ops.BILINEAR, (0, 0, 0), precise=True))
        for i in range(3):
            assert np.array_equal(got[i], O.affine_bilinear(a[i], (w, h), mm, fill=(0, 0, 0))), ("narrow", mm, i)
    monkeypatch.delenv("IMGXF_AFFINE_MF_NARROW")
    # every other frame of a larger batch (frame stride = 2 frames)
    big = dev(np.stack([synth(740 + i, h, w) for i in range(8)]), device)
    got = host(ops.affine(big[::2], m, (w, h), ops.BILINEAR, (0, 0, 0), precise=True))
    for i in range(4):
        assert np.array_equal(got[i], O.affine_bilinear(host(big[2 * i]), (w, h), m, fill=(0, 0, 0)))
    # different output size than the source
    got = host(ops.affine(t, O.rotate_zoom_matrix(w, h, 20.0, 1.3), (352, 224), ops.BILINEAR, (1, 2, 3), precise=True))
    for i in (0, n - 1):
        assert np.array_equal(got[i], O.affine_bilinear(a[i], (352, 224), O.rotate_zoom_matrix(w, h, 20.0, 1.3), fill=(1, 2, 3)))


def test_batched_bilinear_flat_and_extreme_images(device):
    """Flat regions give exact integers (value guard: flat supports are exact in fp32), saturated
    images exercise v = 255."""
    from imagetransformations_amd import ops
    h, w = 192, 256
    flat = np.full((h, w, 3), 200, np.uint8)
    flat[:, ::2] = 255
    steps = (np.arange(h * w * 3, dtype=np.int64).reshape(h, w, 3) // 97 % 2 * 255).astype(np.uint8)
    a = np.stack([flat, steps, np.zeros((h, w, 3), np.uint8), np.full((h, w, 3), 255, np.uint8)])
    for m in (O.rotate_zoom_matrix(w, h, 30.0, 1.5), O.rotate_zoom_matrix(w, h, 0.0, 2.0), O.rotate_zoom_matrix(w, h, 45.0, 1.0)):
        got = host(ops.affine(dev(a, device), m, (w, h), ops.BILINEAR, (0, 0, 0), precise=True))
        for i in range(len(a)):
            assert np.array_equal(got[i], O.affine_bilinear(a[i], (w, h), m, fill=(0, 0, 0))), (m, i)


def test_batched_bilinear_fp32_mode_within_tolerance(device):
    from imagetransformations_amd import ops
    n, h, w = 3, 270, 480
    a = np.stack([synth(760 + i, h, w) for i in range(n)])
    m = O.rotate_zoom_matrix(w, h, 30.0, 1.5)
    got = host(ops.affine(dev(a, device), m, (w, h), ops.BILINEAR, (0, 0, 0), precise=False))
    for i in range(n):
        ref_f, ok = O.affine_bilinear(a[i], (w, h), m, fill=(0, 0, 0), return_float=True)
        ref = np.asarray(ref_f, np.float64)
        tol = 1e-5 * np.maximum(np.abs(ref), 1.0)
        q = lambda v: np.clip(v, 0, 255).astype(np.int64).astype(np.uint8)
        bad = got[i] != q(ref)
        assert ((got[i] == q(ref - tol)) | (got[i] == q(ref + tol)))[bad].all() and bad.mean() < 1e-3


def test_batched_bilinear_4k_pair(device):
    from imagetransformations_amd import ops
    a = np.stack([synth(12345, 2160, 3840), synth(54321, 2160, 3840)])
    m = O.rotate_zoom_matrix(3840, 2160, 30.0, 1.5)
    got = host(ops.affine(dev(a, device), m, (3840, 2160), ops.BILINEAR, (0, 0, 0), precise=True))
    for i in range(2):
        assert np.array_equal(got[i], O.affine_bilinear(a[i], (3840, 2160), m, fill=(0, 0, 0)))


@pytest.mark.parametrize("n,h,w", [(1, 96, 160), (5, 270, 480), (3, 333, 496), (2, 64, 1024)])
def test_batched_nearest_whole_line_kernel_bit_exact(device, monkeypatch, n, h, w):
    """apply_rotation's NEAREST rotation (/root/reference/transformation.py:198-201) through affine_nearest_wq_kernel
    (128 x 16 tiles, wave-private boxes, whole-line stores; 16-byte aligned rows): every frame equals the oracle, for one
    frame per workgroup (default) and for frame loops with the double-buffered boxes and the counted wait."""
    from imagetransformations_amd import ops
    a = np.stack([synth(760 + i, h, w) for i in range(n)])
    t = dev(a, device)
    for angle in (22.5, -7.0, 45.0, 90.0, 135.0, 181.0, 0.0):
        want = np.stack([O.apply_rotation(a[i], -angle) for i in range(n)])         # = Image.rotate(angle)
        for fpb in (None, "2", "3", "24"):
            if fpb: monkeypatch.setenv("IMGXF_AFFINE_FPB", fpb)
            assert np.array_equal(host(ops.rotate(t, angle)), want), (angle, fpb)
            if fpb: monkeypatch.delenv("IMGXF_AFFINE_FPB")
    # a non-black fill and every other frame of a larger batch
    m = O.rotate_plan(w, h, 30.0)[1]
    got = host(ops.affine(t[::2], m, (w, h), ops.NEAREST, (9, 8, 7)))
    for i in range(got.shape[0]):
        assert np.array_equal(got[i], O.affine_nearest(a[2 * i], (w, h), m, fill=(9, 8, 7)))
