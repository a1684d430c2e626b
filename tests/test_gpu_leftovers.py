"""GPU: round-3 fast paths for two round-1 leftovers — whole-pixel NEAREST translations through
the one-pass translation kernel (AugMix translate_x / translate_y, /root/reference/fall_2025/AugMix.py:34-35) and the
3 -> 3 channel permutations on 16-byte chunks (cv2.cvtColor RGB2BGR, /root/reference/transformation.py:233,252) —
against Pillow / NumPy and against the general kernels behind IMGXF_NO_FAST_LEFTOVERS."""
import itertools

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import synth
from test_gpu_parity import dev, host

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hw", [(48, 64), (37, 61), (270, 480), (96, 352)])
def test_whole_pixel_nearest_translations_equal_pillow(device, hw, monkeypatch):
    from imagetransformations_amd import ops
    h, w = hw
    a = np.stack([synth(400 + i, h, w) for i in range(3)])
    t = dev(a, device)
    for tx, ty, fill in [(6, 0, None), (0, 10, None), (-4, 3, (9, 8, 7)), (w + 5, 0, (1, 2, 3)), (0, -h, None), (2.0, -6.0, None), (0, 0, None),
                         (1.5, 0, None)]:                                       # the last one is not a whole pixel: table-driven path
        m = (1, 0, tx, 0, 1, ty)
        got = host(ops.affine(t, m, (w, h), ops.NEAREST, fill))
        for i in range(3):
            want = np.asarray(Image.fromarray(a[i]).transform((w, h), Image.AFFINE, m, fillcolor=fill))
            assert np.array_equal(got[i], want), (m, fill, i)
        monkeypatch.setenv("IMGXF_NO_FAST_LEFTOVERS", "1")
        assert np.array_equal(host(ops.affine(t, m, (w, h), ops.NEAREST, fill)), got), m
        monkeypatch.delenv("IMGXF_NO_FAST_LEFTOVERS")


@pytest.mark.parametrize("hw", [(32, 32), (37, 61), (64, 48), (270, 480), (5, 16)])
def test_channel_permutations_every_order(device, hw, monkeypatch):
    from imagetransformations_amd import ops
    a = np.stack([synth(420 + i, *hw) for i in range(2)])
    t = dev(a, device)
    for perm in itertools.product(range(3), repeat=3):                          # all 27 maps, duplicates included
        got = host(ops.permute_channels(t, perm))
        assert np.array_equal(got, a[..., list(perm)]), perm
    monkeypatch.setenv("IMGXF_NO_FAST_LEFTOVERS", "1")
    assert np.array_equal(host(ops.permute_channels(t, (2, 1, 0))), a[..., ::-1])
    monkeypatch.delenv("IMGXF_NO_FAST_LEFTOVERS")
    big = dev(np.stack([synth(440 + i, 64, 96) for i in range(4)]), device)     # a strided view: every other frame, a row window
    view = big[::2, 8:40]
    assert np.array_equal(host(ops.permute_channels(view, (2, 1, 0))), host(view)[..., ::-1])


@pytest.mark.parametrize("hw", [(16, 128), (128, 16), (37, 61), (270, 480), (2, 2), (131, 257), (5, 300)])
def test_quarter_turns_batches_and_views_equal_numpy(device, hw):
    """Image.transpose(ROTATE_90 / ROTATE_270) (the fast paths of Image.rotate, apply_rotation's right angles,
    /root/reference/transformation.py:198-201): batches, partial tiles on every side, strided views; equal to np.rot90."""
    from imagetransformations_amd import ops
    h, w = hw
    a = np.stack([synth(460 + i, h, w) for i in range(3)])
    t = dev(a, device)
    for k in (1, 3):
        assert np.array_equal(host(ops.rot90(t, k)), np.rot90(a, k, axes=(1, 2))), k
    if h >= 8 and w >= 12:
        view = t[::2, 2:h - 1, 3:w - 2]                                         # rows at odd byte offsets, frame stride of two frames
        for k in (1, 3):
            assert np.array_equal(host(ops.rot90(view, k)), np.rot90(host(view), k, axes=(1, 2))), ("view", k)


@pytest.mark.parametrize("hw", [(48, 64), (37, 60), (270, 480)])
def test_nearest_zoom_dword_gather_equals_pillow(device, hw):
    """NEAREST pure scale / translate (ImagingScaleAffine; camera-distance style zooms): the packed-RGB kernel reads one
    unaligned dword per source pixel, the row's last pixel included."""
    from imagetransformations_amd import ops
    h, w = hw
    a = np.stack([synth(480 + i, h, w) for i in range(2)])
    t = dev(a, device)
    for m in [(1 / 1.2, 0, w * 0.1 / 1.2, 0, 1 / 1.2, h * 0.1 / 1.2), (0.5, 0, 0, 0, 0.5, 0), (1.0, 0, 0.25, 0, 1.0, 0.75),
              (1.7, 0, -3.2, 0, 0.6, 2.0), (1.0, 0, w - 1.5, 0, 1.0, 0)]:
        got = host(ops.affine(t, m, (w, h), ops.NEAREST, (5, 6, 7)))
        for i in range(2):
            want = np.asarray(Image.fromarray(a[i]).transform((w, h), Image.AFFINE, m, fillcolor=(5, 6, 7)))
            assert np.array_equal(got[i], want), (m, i)
