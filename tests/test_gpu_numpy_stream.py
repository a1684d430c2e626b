"""GPU: NumPy's legacy normal stream on the device (csrc/noise_rng.hip imgxf_mt19937_blocks + numpy_stream.py): the block
kernel against the NumPy restatement, and draw_on_device against np.random.normal itself — numbers and generator state."""
import numpy as np
import pytest
import torch

from imagetransformations_amd import _ffi as F, numpy_stream as NS

pytestmark = pytest.mark.gpu


def test_mt19937_block_kernel_equals_the_restatement(device):
    key = np.random.RandomState(77).get_state()[1]
    nblocks = 40
    key_d = torch.from_numpy(key.astype(np.uint32).view(np.int32).copy()).to(device)
    raw = torch.empty(((nblocks + 1) * 624,), dtype=torch.int32, device=device)
    F.call("imgxf_mt19937_blocks", key_d.data_ptr(), raw.data_ptr(), nblocks, torch.cuda.current_stream().cuda_stream)
    got = raw.cpu().numpy().view(np.uint32).reshape(nblocks + 1, 624)
    want = key.astype(np.uint32)
    for b in range(nblocks + 1):
        assert np.array_equal(got[b], want), b
        want = NS.mt_next_block(want)


@pytest.mark.parametrize("seed,requests", [(0, [(375 * 500 * 3, 0.05 * 255)]), (5, [(999, 2.0), (1, 1.0), (64 * 48 * 3, 25.5), (7, 0.1)]),
                                           (9, [(32 * 32 * 3, 0.08 * 255)] * 9)])
def test_draw_on_device_is_np_random_normal(device, seed, requests):
    np.random.seed(seed)
    np.random.random_sample(seed)                      # somewhere inside a block
    st = np.random.get_state()
    want = [np.random.normal(0, s, n).astype(np.float32) for n, s in requests]
    after = np.random.get_state()
    np.random.set_state(st)
    got = NS.draw_on_device(requests, device)
    assert got is not None
    for g, w in zip(got, want):
        assert np.array_equal(g.cpu().numpy(), w)
    now = np.random.get_state()
    assert now[2] == after[2] and np.array_equal(now[1], after[1]) and now[3] == after[3] and (now[4] == after[4] or not now[3])
    follow = np.random.normal(0, 1, 5)                 # the host's next draw continues where the device stopped
    np.random.set_state(after)
    assert np.array_equal(follow, np.random.normal(0, 1, 5))


def test_drivers_give_the_same_images_with_the_noise_drawn_on_the_device(device, monkeypatch):
    """apply_gaussian_noise and the batched driver with the default mode (NumPy's stream computed on the device) against the
    same calls with the host drawing the numbers (IMGXF_NOISE_RNG=numpy-host): identical pixels, identical np.random state."""
    import random
    from PIL import Image
    from conftest import synth
    from imagetransformations_amd import transformation as T
    imgs = [(Image.fromarray(synth(900 + i, 61 + 3 * (i % 2), 83)), f"img_{i}.jpeg") for i in range(5)]
    outs, states = [], []
    for mode in ("numpy", "numpy-host"):
        monkeypatch.setattr(T, "NOISE_RNG", mode)
        random.seed(3); np.random.seed(3)
        a = T.apply_gaussian_noise(imgs[0][0], 0.07)
        b = T.apply_all_transformations_batched(imgs)
        outs.append([np.asarray(a)] + [np.asarray(im) for im in b])
        states.append(np.random.get_state())
    for x, y in zip(*outs):
        assert np.array_equal(x, y)
    assert states[0][2] == states[1][2] and np.array_equal(states[0][1], states[1][1]) and states[0][3] == states[1][3]


def test_pool_gaussian_noise_doubles_on_the_device(device, monkeypatch):
    """TransformationPool.gaussian_noise adds float64 noise (cifar_image_transformations.py:39-48): the device-computed doubles
    give the pixels of the host draw and leave np.random where it would be, for an ImageNet-size image (CIFAR sizes stay on
    the host)."""
    from PIL import Image
    from conftest import synth
    from imagetransformations_amd import transformation as T
    from imagetransformations_amd.pool import TransformationPool as P
    img = Image.fromarray(synth(950, 180, 240))
    res = []
    for mode in ("numpy", "numpy-host"):
        monkeypatch.setattr(T, "NOISE_RNG", mode)
        np.random.seed(17)
        a = np.asarray(P.gaussian_noise(img, 3))
        b = np.asarray(P.gaussian_noise(img, 5))
        res.append((a, b, np.random.get_state()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    s0, s1 = res[0][2], res[1][2]
    assert s0[2] == s1[2] and np.array_equal(s0[1], s1[1]) and s0[3] == s1[3] and (s0[4] == s1[4] or not s0[3])


def test_passes_over_the_stream_hand_the_state_on(device, monkeypatch):
    """Requests beyond PASS_NORMALS are served by several passes; np.random's state carries between them."""
    monkeypatch.setattr(NS, "PASS_NORMALS", 5000)
    requests = [(3001, 2.0), (1999, 1.0), (4000, 3.0), (7, 1.0), (9001, 0.5)]
    np.random.seed(31)
    want = [np.random.normal(0, s, n).astype(np.float32) for n, s in requests]
    after = np.random.get_state()
    np.random.seed(31)
    got = NS.draw_on_device(requests, device)
    for g, w in zip(got, want):
        assert np.array_equal(g.cpu().numpy(), w)
    now = np.random.get_state()
    assert now[2] == after[2] and np.array_equal(now[1], after[1]) and now[3] == after[3] and (now[4] == after[4] or not now[3])


def test_jump_ahead_stretches_equal_the_sequential_stream(device):
    """The state sequence generated in stretches (imgxf_mt19937_jump + imgxf_mt19937_stretches, stride = 624 * 2^16 words from
    mt19937_jump.npz) is word for word the sequential one — across three stretch boundaries, for a key in the middle of its life."""
    tables = NS._jump_tables(torch.device(device))
    assert tables is not None, "the jump polynomial failed its self-check"
    bps = tables["bps"]
    rs = np.random.RandomState(4242)
    rs.random_sample(100000)                              # a state some blocks into its stream
    key = rs.get_state()[1].astype(np.uint32)
    key_d = torch.from_numpy(key.view(np.int32).copy()).to(device)
    nblocks = 3 * bps + 1234
    st = torch.cuda.current_stream().cuda_stream
    par = NS.generate_stream(key_d, nblocks, torch.device(device), st)
    seq = torch.empty(((nblocks + 1) * 624,), dtype=torch.int32, device=device)
    F.call("imgxf_mt19937_blocks", key_d.data_ptr(), seq.data_ptr(), nblocks, st)
    assert torch.equal(par, seq)


def test_a_draw_longer_than_a_stretch_is_np_random_normal(device):
    """64 ImageNet-size images' worth of normals in one call: the stream comes from several stretches."""
    n = 375 * 500 * 3
    requests = [(n, 0.05 * 255 + i) for i in range(64)]
    np.random.seed(123)
    st = np.random.get_state()
    want_first = np.random.normal(0, requests[0][1], n).astype(np.float32)
    for _, s in requests[1:-1]:
        np.random.normal(0, s, n)
    want_last = np.random.normal(0, requests[-1][1], n).astype(np.float32)
    after = np.random.get_state()
    np.random.set_state(st)
    got = NS.draw_on_device(requests, device)
    assert np.array_equal(got[0].cpu().numpy(), want_first) and np.array_equal(got[-1].cpu().numpy(), want_last)
    now = np.random.get_state()
    assert now[2] == after[2] and np.array_equal(now[1], after[1]) and now[3] == after[3] and (now[4] == after[4] or not now[3])
