"""GPU: the opt-in device RNG of the noise transform (IMGXF_NOISE_RNG=device; csrc/noise_rng.hip).
/root/reference/transformation.py:272-281 draws np.random.normal on the host; the device path is a DIFFERENT stream
(Philox4x32-10 + Box-Muller), so parity is at the level of the distribution: the raw uint32 stream
is pinned bit for bit (Random123 known answers through the NumPy oracle), the noisy pixels by their moments, a
Kolmogorov-Smirnov bound against the exact law of clip(trunc(p + N(0, sigma))), and the clipping at both ends."""
import math

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import synth
from oracle import philox_oracle as PO

pytestmark = pytest.mark.gpu


def test_device_uint32_stream_equals_philox4x32_10(device):
    from imagetransformations_amd import ops
    for seed, offset, count in [(0, 0, 16), (0x0123456789abcdef, 0, 4096), (2 ** 63 - 5, 4 * 12345, 1024), (7, 2 ** 34, 64)]:
        got = ops.philox_u32(count, seed, offset, device).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, PO.stream(count, seed, offset)), (seed, offset)
    assert [int(v) for v in ops.philox_u32(4, 0, 0, device).cpu().numpy().view(np.uint32)] == list(PO.KAT[0][2])


def _phi(x):
    return 0.5 * (1.0 + math.erf(x / math.sqrt(2.0)))


@pytest.mark.parametrize("base,sigma", [(128, 12.75), (128, 2.55), (40, 25.5), (3, 5.0), (250, 7.0)])
def test_noise_distribution(device, base, sigma):
    from imagetransformations_amd import ops
    h, w = 512, 768
    t = torch.full((2, h, w, 3), base, dtype=torch.uint8, device=device)
    out = ops.add_noise_device(t, sigma, seed=1234567).cpu().numpy().astype(np.int64).reshape(-1)
    n = out.size
    # exact law of clip(trunc(base + z), 0, 255), z ~ N(0, sigma): P(out <= k) = Phi((k + 1 - base) / sigma) for 0 <= k < 255
    ks = np.arange(0, 255)
    want_cdf = np.array([_phi((k + 1 - base) / sigma) for k in ks])
    got_cdf = np.cumsum(np.bincount(out, minlength=256))[:255] / n
    assert np.abs(got_cdf - want_cdf).max() < 2.5 / math.sqrt(n) + 2e-4, np.abs(got_cdf - want_cdf).max()
    if 4 * sigma < base < 255 - 4 * sigma:                                  # no clipping: moments of trunc(base + z)
        assert abs(out.mean() - (base - 0.5)) < 5 * sigma / math.sqrt(n) + 1e-3
        assert abs(out.std() - math.sqrt(sigma ** 2 + 1 / 12)) < 0.01 * sigma + 5e-3
    assert out.min() >= 0 and out.max() <= 255


def test_noise_is_counter_based_and_seeded(device):
    from imagetransformations_amd import ops
    a = np.stack([synth(900 + i, 37, 64) for i in range(3)])                # 192-byte rows
    t = torch.from_numpy(a).to(device)
    x = ops.add_noise_device(t, 10.0, seed=42)
    assert torch.equal(x, ops.add_noise_device(t, 10.0, seed=42))
    assert not torch.equal(x, ops.add_noise_device(t, 10.0, seed=43))
    per_frame = 37 * 64 * 3
    for f in range(3):                                                      # a frame alone, numbered by `offset`
        assert torch.equal(x[f], ops.add_noise_device(t[f], 10.0, seed=42, offset=f * per_frame))
    assert torch.equal(ops.add_noise_device(t, 0.0, seed=1), t)             # sigma 0: clip(f32(p)) == p
    # independent channels / neighbours: correlation of the noise of adjacent bytes is that of independent draws
    flat = torch.full((1, 256, 768, 3), 128, dtype=torch.uint8, device=device)
    z = ops.add_noise_device(flat, 20.0, seed=5).cpu().numpy().astype(np.float64).reshape(-1) - 127.5
    for lag in (1, 2, 3, 4, 768 * 3):
        c = float(np.corrcoef(z[:-lag], z[lag:])[0, 1])
        assert abs(c) < 5.0 / math.sqrt(z.size), (lag, c)
    odd = torch.from_numpy(synth(950, 5, 7)).to(device)                     # 21-byte rows: the byte-wise tail path
    y = ops.add_noise_device(odd, 3.0, seed=9)
    assert y.shape == odd.shape and int((y.int() - odd.int()).abs().max()) <= 20


def test_facade_opt_in_keeps_seeded_runs_repeatable(device, monkeypatch):
    from imagetransformations_amd import transformation as T
    img = Image.fromarray(synth(960, 48, 64))
    monkeypatch.setattr(T, "NOISE_RNG", "device")
    np.random.seed(3); a = np.asarray(T.apply_gaussian_noise(img, 0.05))
    np.random.seed(3); b = np.asarray(T.apply_gaussian_noise(img, 0.05))
    np.random.seed(4); c = np.asarray(T.apply_gaussian_noise(img, 0.05))
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    d = a.astype(np.float64) - np.asarray(img)
    assert abs(d.std() - 0.05 * 255) < 1.5                                  # (clipping at 0 / 255 trims the tails a little)
    # the batched driver draws its seeds at the same points of the np.random stream as the per-image loop
    import random
    imgs = [(Image.fromarray(synth(970 + i, 48, 64)), f"/x/im{i}.JPEG") for i in range(3)]
    monkeypatch.setattr(T, "output_dir", None)
    random.seed(11); np.random.seed(11); one = T.apply_all_transformations_per_image(imgs)
    random.seed(11); np.random.seed(11); many = T.apply_all_transformations_batched(imgs)
    assert len(one) == len(many) == 24
    for p, q in zip(one, many):
        assert np.array_equal(np.asarray(p), np.asarray(q))
    monkeypatch.setattr(T, "NOISE_RNG", "numpy")
    np.random.seed(3)
    ref = np.asarray(T.apply_gaussian_noise(img, 0.05))
    np.random.seed(3)
    noise = np.random.normal(0, 0.05 * 255, (48, 64, 3)).astype(np.float32)
    assert np.array_equal(ref, np.clip(np.asarray(img).astype(np.float32) + noise, 0, 255).astype(np.uint8))   # default = the reference's stream
