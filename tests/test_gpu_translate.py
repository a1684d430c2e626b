"""apply_translation (/root/reference/transformation.py:284-307: black canvas + crop + paste) as one kernel,
and the destination-aligned rectangle copy behind Image.crop / Image.paste: bit-exact against the oracle for
every shift direction, shifts past the image, channel counts, strided views and padded destinations."""
import numpy as np
import pytest
import torch

from conftest import synth
from oracle import imgxf_oracle as O
from test_gpu_parity import dev, host

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hw", [(32, 32), (37, 61), (270, 480), (65, 1283)])
def test_translation_bit_exact(device, hw):
    from imagetransformations_amd import ops
    from imagetransformations_amd import transformation as T
    a = synth(300, *hw)
    t = dev(a, device)
    h, w = hw
    for tx, ty in [(0, 0), (5, 0), (-5, 0), (0, 7), (0, -7), (50, -50), (-45.7, 33.2), (w - 1, h - 1), (-(w - 1), 3),
                   (w, 0), (0, -h), (3 * w, 2), (1, 1), (-16, -16), (16, 16)]:
        assert np.array_equal(host(T._translation_t(t, tx, ty)), O.apply_translation(a, tx, ty)), (tx, ty)
    got = host(ops.translate(t, 4, -3, (9, 8, 7)))
    want = np.empty_like(a); want[:] = (9, 8, 7)
    want[:h - 3, 4:] = a[3:, :w - 4]
    assert np.array_equal(got, want)


@pytest.mark.parametrize("c", [1, 4])
def test_translation_other_channel_counts_batches_and_views(device, c):
    from imagetransformations_amd import ops
    rng = np.random.default_rng(c)
    a = rng.integers(0, 256, (5, 70, 131, c), dtype=np.uint8)
    t = torch.from_numpy(a).to(device)
    for dx, dy in [(7, -9), (-130, 69), (0, 0)]:
        got = ops.translate(t, dx, dy, tuple(range(1, c + 1))).cpu().numpy()
        want = np.empty_like(a); want[:] = np.arange(1, c + 1, dtype=np.uint8)
        ys0, ys1, xs0, xs1 = max(0, -dy), min(70, 70 - dy), max(0, -dx), min(131, 131 - dx)
        if ys0 < ys1 and xs0 < xs1:
            want[:, ys0 + dy:ys1 + dy, xs0 + dx:xs1 + dx] = a[:, ys0:ys1, xs0:xs1]
        assert np.array_equal(got, want), (dx, dy)
    big = torch.from_numpy(rng.integers(0, 256, (6, 80, 150, c), dtype=np.uint8)).to(device)
    view = big[::2, 5:75, 10:141]                                   # strided frames and rows, unaligned base
    assert torch.equal(ops.translate(view, 3, 4), ops.translate(view.contiguous(), 3, 4))


def test_crop_and_paste_at_every_alignment(device):
    """copy_rect cuts its chunks on the destination's 16-byte grid; every (source, destination) byte phase."""
    from imagetransformations_amd import ops
    a = synth(310, 40, 200)
    t = dev(a, device)
    for sx in range(0, 7):
        for dx in range(0, 6):
            for rw in (1, 5, 17, 64, 150):
                canvas = torch.full((40, 230, 3), 77, dtype=torch.uint8, device=device)
                ops.copy_rect(t, canvas, sx, 3, dx, 2, rw, 30)
                want = np.full((40, 230, 3), 77, np.uint8)
                want[2:32, dx:dx + rw] = a[3:33, sx:sx + rw]
                assert np.array_equal(canvas.cpu().numpy(), want), (sx, dx, rw)


@pytest.mark.parametrize("hw", [(32, 32), (37, 61), (270, 480), (65, 1284), (5, 4), (64, 1283)])
def test_mirrors_bit_exact(device, hw):
    """FLIP_LEFT_RIGHT (vert_flip of fall_2025/transformations_code:39-41), FLIP_TOP_BOTTOM and ROTATE_180: the
    dword kernels (widths that are multiples of 4) and the per-pixel fallbacks, batches and strided views."""
    from imagetransformations_amd import ops
    a = np.stack([synth(320 + i, *hw) for i in range(3)])
    t = dev(a, device)
    assert np.array_equal(host(ops.flip(t)), a[:, :, ::-1])
    assert np.array_equal(host(ops.flip(t, top_bottom=True)), a[:, ::-1])
    assert np.array_equal(host(ops.rot90(t, 2)), a[:, ::-1, ::-1])
    h, w = hw
    if w >= 12 and h >= 6:
        view = t[::2, 1:h - 2, 4:w - 4]                               # strided frames, offset rows, width still a multiple of 4 or not
        ref = view.contiguous()
        assert torch.equal(ops.flip(view), ops.flip(ref)) and torch.equal(ops.rot90(view, 2), ops.rot90(ref, 2))
        assert torch.equal(ops.flip(view, top_bottom=True), ops.flip(ref, top_bottom=True))
    g = torch.from_numpy(np.ascontiguousarray(a[..., :1])).to(device)  # gray [N,H,W,1]: per-pixel path for left-right
    assert np.array_equal(ops.flip(g).cpu().numpy(), a[..., :1][:, :, ::-1])


@pytest.mark.parametrize("hw", [(32, 32), (37, 61), (270, 480), (65, 130), (1, 9), (9, 1), (33, 31)])
def test_quarter_turns_bit_exact(device, hw):
    """Image.transpose(ROTATE_90 / ROTATE_270) through the LDS tile, ragged sizes, batches, gray and RGBA."""
    from imagetransformations_amd import ops
    a = np.stack([synth(330 + i, *hw) for i in range(3)])
    t = dev(a, device)
    for k in (1, 3):
        assert np.array_equal(host(ops.rot90(t, k)), np.rot90(a, k, axes=(1, 2))), k
    rng = np.random.default_rng(1)
    for c in (1, 4):
        b = rng.integers(0, 256, (2, hw[0], hw[1], c), dtype=np.uint8)
        for k in (1, 3):
            assert np.array_equal(ops.rot90(torch.from_numpy(b).to(device), k).cpu().numpy(), np.rot90(b, k, axes=(1, 2)))


@pytest.mark.parametrize("m", [(1, 0, -3.5, 0, 1, 2.25), (0.5, 0, 0, 0, 0.5, 0), (2.3, 0, -10, 0, 1.7, 5), (-1, 0, 60, 0, 1, 0),
                               (1, 0, 1000, 0, 1, 0), (1.0, 0, 0.4999, 0, 1.0, -0.5)])
def test_axis_aligned_nearest_dword_kernel(device, m):
    """ImagingScaleAffine (AugMix translate_x/y, zoom without rotation) with 4 pixels per lane: widths that are
    multiples of 4, batches, fill colour; the ragged widths keep the per-pixel kernel (test_gpu_parity)."""
    from imagetransformations_amd import ops
    a = np.stack([synth(340 + i, 37, 64) for i in range(3)])
    t = dev(a, device)
    for size in [(64, 37), (128, 50), (20, 8)]:
        got = host(ops.affine(t, m, size, ops.NEAREST, (9, 8, 7)))
        for i in range(3):
            assert np.array_equal(got[i], O.affine_nearest(a[i], size, [float(v) for v in m], fill=(9, 8, 7))), (m, size, i)
