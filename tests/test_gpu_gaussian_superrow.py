"""Super-row strips of the marching Gaussian (csrc/sepconv_march.inc): the same row of G frames is
cut into 1 KiB strips as one run, so frame seams fall inside waves.  Batches whose size is not
a multiple of G, strided frame views, the fp32 side output, the fixed-point instances and all
channel counts must equal the per-frame oracle (cv2.GaussianBlur, float definition:
/root/reference/transformation.py:249)."""
import numpy as np
import pytest
import torch

from conftest import synth
from oracle import imgxf_oracle as O
from test_gpu_parity import assert_quantised_close, dev, host

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,h,w,c", [(9, 40, 480, 3), (8, 33, 1920, 3), (5, 24, 3840, 3), (4, 21, 1360, 1), (11, 19, 272, 4),
                                     (3, 18, 352, 3), (2, 17, 5120, 3)])
@pytest.mark.parametrize("radius", [5 / 6, 1.0])
def test_superrow_batches_match_per_frame_oracle(device, n, h, w, c, radius):
    from imagetransformations_amd import ops
    a = np.stack([synth(100 + i, h, w, c) for i in range(n)])
    if c == 1:
        a = a[..., None]
    k = O.blur_ksize(radius)
    out, f32 = ops.gaussian_blur(dev(a, device), k, radius, return_f32=True)
    out, f32 = host(out), host(f32)
    for i in range(n):
        src = a[i, ..., 0] if c == 1 else a[i]
        ref = O.gaussian_blur_f64(src, k, radius)
        if c == 1:
            ref = ref[..., None]
        assert_quantised_close(out[i], f32[i], ref, O.saturate_u8)
    # the u8-only launch (no fp32 side output) must give the same bytes
    assert np.array_equal(host(ops.gaussian_blur(dev(a, device), k, radius)), out)


def test_superrow_strided_frames_and_padded_rows(device):
    from imagetransformations_amd import ops
    big = dev(np.stack([synth(200 + i, 30, 480) for i in range(14)]), device)
    sub = big[::2]                                   # frame stride = 2 frames
    got = host(ops.gaussian_blur(sub, 5, 5 / 6))
    for i in range(7):
        assert np.array_equal(got[i], host(ops.gaussian_blur(big[2 * i], 5, 5 / 6)))
    wide = dev(np.stack([synth(300 + i, 30, 496) for i in range(8)]), device)
    win = wide[:, :, :480]                           # row stride 1488 B, rows of 1440 B
    got = host(ops.gaussian_blur(win, 5, 5 / 6))
    ref = host(ops.gaussian_blur(win.contiguous(), 5, 5 / 6))
    assert np.array_equal(got, ref)
    for i in (0, 7):
        want = O.gaussian_blur(host(win[i]), 5, 5 / 6)
        assert np.abs(got[i].astype(int) - want.astype(int)).max() <= 1 and (got[i] != want).mean() < 1e-3


def test_superrow_fixed_point_is_bit_exact(device):
    from imagetransformations_amd import ops
    a = np.stack([synth(400 + i, 26, 480) for i in range(9)])
    for radius in (5 / 6, 1.5):
        k = O.blur_ksize(radius)
        got = host(ops.gaussian_blur(dev(a, device), k, radius, fixed_point=True))
        for i in range(9):
            assert np.array_equal(got[i], O.gaussian_blur_cv_fixed(a[i], k, radius))


def test_superrow_group_size_does_not_change_results(device, monkeypatch):
    from imagetransformations_amd import ops
    a = dev(np.stack([synth(500 + i, 50, 3840) for i in range(8)]), device)
    base = host(ops.gaussian_blur(a, 5, 5 / 6))
    for g in ("1", "2", "3", "8"):
        monkeypatch.setenv("IMGXF_MARCH_GROUP", g)
        assert np.array_equal(host(ops.gaussian_blur(a, 5, 5 / 6)), base), g
    monkeypatch.delenv("IMGXF_MARCH_GROUP")
    for spb in ("1", "2", "4"):
        monkeypatch.setenv("IMGXF_MARCH_SPB", spb)
        assert np.array_equal(host(ops.gaussian_blur(a, 5, 5 / 6)), base), spb
