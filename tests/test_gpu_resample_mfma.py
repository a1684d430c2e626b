"""The fused matrix-core resample (csrc/resample_mfma.inc) behind img.resize(size, LANCZOS)
(/root/reference/transformation.py:179) and apply_scale's resize + centre crop (:182-187):
bit-exact against the oracle at sizes it finishes quickly, byte-identical to the two-pass
vector kernels at full size, for every filter, channel count, ragged width and strided view."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import synth
from oracle import imgxf_oracle as O
from test_gpu_parity import dev, host

pytestmark = pytest.mark.gpu


def _ksteps(in_hw, out_hw, c, resample=1, window=None):
    """k-steps of the fused kernel's plan for this geometry (0 = the two-pass kernels run)."""
    from imagetransformations_amd import _ffi as F
    from imagetransformations_amd import ops
    plan = ops._plans.get(in_hw[0], in_hw[1], out_hw[0], out_hw[1], c, 0, resample, window)
    k = ctypes.c_int(-1)
    F.call("imgxf_resample_plan_kernel", plan, ctypes.byref(k))
    return k.value


@pytest.mark.parametrize("hw", [(64, 96), (135, 240), (270, 480), (334, 500), (97, 131)])
@pytest.mark.parametrize("s", [0.5, 0.9, 1.1, 1.2000000000000002, 1.5, 2.0])
def test_fused_resample_bit_exact(device, hw, s):
    from imagetransformations_amd import ops
    a = synth(31, *hw)
    h, w = hw
    nw, nh = int(w * s), int(h * s)
    out = host(ops.resize_lanczos(dev(a, device), (nw, nh)))
    assert np.array_equal(out, O.resize_lanczos(a, (nw, nh)))
    if (w * 3) % 4 == 0 and s >= 0.9:            # the reference's grid is 0.9 .. 1.4; 0.5x needs 4 k-steps -> two-pass
        assert _ksteps(hw, (nh, nw), 3) > 0, "the matrix-core plan should cover this geometry"


@pytest.mark.parametrize("resample", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("c", [1, 3, 4])
def test_fused_resample_filters_and_channels(device, resample, c):
    from imagetransformations_amd import ops
    rng = np.random.default_rng(100 * resample + c)
    a = rng.integers(0, 256, (3, 120, 176, c), dtype=np.uint8)
    a[0, :40] = 255; a[1, :, 60:90] = 0          # saturating edges: negative lobes clip both ways
    t = torch.from_numpy(a).to(device)
    for size in [(200, 150), (130, 99), (176, 150), (352, 61)]:
        out = ops.resize(t, size, resample).cpu().numpy()
        for i in range(3):
            ref = O.resize(a[i] if c > 1 else a[i, :, :, 0], size, resample)
            got = out[i] if c > 1 else out[i, :, :, 0]
            assert np.array_equal(got, ref), (resample, c, size)
    assert _ksteps((120, 176), (150, 200), c, resample) > 0


def test_fused_resample_window_equals_resize_then_crop(device):
    """apply_scale's crop window: only the window's rows / columns are produced."""
    from imagetransformations_amd import ops
    a = np.stack([synth(40 + i, 216, 384) for i in range(2)])
    t = torch.from_numpy(a).to(device)
    for s in (1.1, 1.3, 1.5):
        nw, nh, mode, ox, oy = O.scale_geometry(384, 216, s)
        assert mode == "crop"
        out = ops.resize_crop(t, (nw, nh), (ox, oy, ox + 384, oy + 216)).cpu().numpy()
        for i in range(2):
            assert np.array_equal(out[i], O.apply_scale(a[i], s))
        assert _ksteps((216, 384), (nh, nw), 3, 1, (ox, oy, 384, 216)) > 0


def test_fused_equals_two_pass_at_full_size(device, monkeypatch):
    """4K and 1080p, 1.1x with the centre crop (the ops-table row) and a plain 0.5x / 1.5x resize."""
    from imagetransformations_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    for (h, w, n) in [(2160, 3840, 3), (1080, 1920, 5)]:
        t = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, generator=g).to(device)
        nw, nh, _, ox, oy = O.scale_geometry(w, h, 1.1)
        box = (ox, oy, ox + w, oy + h)
        assert _ksteps((h, w), (nh, nw), 3, 1, (ox, oy, w, h)) > 0
        fused = ops.resize_crop(t, (nw, nh), box)
        plain = [ops.resize_lanczos(t, sz) for sz in [(w // 2, h // 2), (w * 3 // 2, h * 3 // 2)]]
        monkeypatch.setenv("IMGXF_RESAMPLE_NO_MFMA", "1")
        assert torch.equal(fused, ops.resize_crop(t, (nw, nh), box))
        for sz, p in zip([(w // 2, h // 2), (w * 3 // 2, h * 3 // 2)], plain):
            assert torch.equal(p, ops.resize_lanczos(t, sz))
        monkeypatch.delenv("IMGXF_RESAMPLE_NO_MFMA")


def test_fused_resample_strided_views_and_ragged_widths(device):
    from imagetransformations_amd import ops
    big = torch.from_numpy(np.stack([synth(50 + i, 200, 300) for i in range(2)])).to(device)
    src = big[:, 8:168, 20:220]                    # 160 x 200 window: row stride 900, base offset 60 + 7200
    a = src.cpu().numpy()
    canvas = torch.zeros((2, 300, 400, 3), dtype=torch.uint8, device=device)
    dst = canvas[:, 10:186, 32:252]                # 176 x 220
    ops.resize(src, (220, 176), 1, out=dst)
    for i in range(2):
        assert np.array_equal(dst[i].cpu().numpy(), O.resize_lanczos(a[i], (220, 176)))
    assert int(canvas[:, :10].max()) == 0 and int(canvas[:, 186:].max()) == 0
    assert int(canvas[:, :, :32].max()) == 0 and int(canvas[:, :, 252:].max()) == 0
    # widths whose byte rows are not multiples of 16 (the last lanes store byte-wise) or of 4 (two-pass fallback)
    for (h, w, size) in [(64, 100, (77, 50)), (50, 67, (75, 41)), (33, 44, (47, 90))]:
        b = synth(60, h, w)
        assert np.array_equal(host(ops.resize_lanczos(dev(b, device), size)), O.resize_lanczos(b, size)), (h, w, size)


def test_fused_resample_row_chunking(device, monkeypatch):
    """Chunk length is a plan-time knob; every setting gives the same bytes."""
    from imagetransformations_amd import ops
    a = synth(70, 400, 256)
    ref = O.resize_lanczos(a, (300, 470))
    for oc in ("1", "2", "5", "32"):
        monkeypatch.setenv("IMGXF_RESAMPLE_MFMA_OC", oc)
        ops._plans.clear()
        assert np.array_equal(host(ops.resize_lanczos(dev(a, device), (300, 470))), ref), oc
    monkeypatch.delenv("IMGXF_RESAMPLE_MFMA_OC")
    ops._plans.clear()


def test_fused_resample_random_geometries(device):
    """Seeded sweep over sizes, scales (both directions, anisotropic), filters, channel counts and crop windows:
    every result equals the oracle; the fused kernel must have taken a fair share of them."""
    from imagetransformations_amd import ops
    rng = np.random.default_rng(20261004)
    fused = 0
    for it in range(60):
        h, w = int(rng.integers(33, 260)), int(rng.integers(12, 90)) * 4      # byte rows of RGB / gray stay 4-byte multiples
        c = int(rng.choice([1, 3, 3, 3, 4]))
        sx, sy = float(rng.uniform(0.62, 2.2)), float(rng.uniform(0.62, 2.2))
        nw, nh = max(1, int(w * sx)), max(1, int(h * sy))
        resample = int(rng.choice([1, 1, 2, 3, 4, 5]))
        a = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
        if it % 5 == 0: a[: h // 2] = 255                                      # saturation under negative lobes
        t = torch.from_numpy(a).to(device)
        ref = O.resize(a if c > 1 else a[:, :, 0], (nw, nh), resample)
        if it % 3 == 0 and nw > 8 and nh > 8 and nw != w and nh != h:
            l, tp = int(rng.integers(0, nw // 2)), int(rng.integers(0, nh // 2))
            r, b = int(rng.integers(l + 1, nw + 1)), int(rng.integers(tp + 1, nh + 1))
            got = ops.resize_crop(t, (nw, nh), (l, tp, r, b), resample).cpu().numpy()
            want = ref[tp:b, l:r]
            fused += _ksteps((h, w), (nh, nw), c, resample, (l, tp, r - l, b - tp)) > 0
        else:
            got = ops.resize(t, (nw, nh), resample).cpu().numpy()
            want = ref
            fused += _ksteps((h, w), (nh, nw), c, resample) > 0
        got = got if c > 1 else got[:, :, 0]
        assert np.array_equal(got, want), (it, h, w, c, nw, nh, resample)
    assert fused >= 30
