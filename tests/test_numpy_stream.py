"""NumPy's legacy normal stream restated for the device (imagetransformations_amd/numpy_stream.py): on the CPU, with the
MT19937 blocks from the NumPy restatement `mt_next_block`, the torch evaluation must give np.random.normal's numbers and
leave the generator in np.random's state — for chains of draws, odd counts (the cached second normal), states in the
middle of a block and at its end."""
import numpy as np
import pytest
import torch

from imagetransformations_amd import numpy_stream as NS


def raw_stream(key, nblocks):
    blocks = [key.astype(np.uint32)]
    for _ in range(nblocks):
        blocks.append(NS.mt_next_block(blocks[-1]))
    return torch.from_numpy(np.concatenate(blocks).astype(np.uint32).view(np.int32).copy())     # uint32 bit patterns, as the kernel writes them


def test_mt_block_restatement_is_numpys_generator():
    rs = np.random.RandomState(12345)
    _, key, pos, _, _ = rs.get_state()
    assert pos == 624
    nxt = NS.mt_next_block(key)
    words = rs.randint(0, 2 ** 32, size=624, dtype=np.uint64).astype(np.uint32)      # the tempered words of the next block
    assert np.array_equal(NS.temper(torch.from_numpy(nxt.astype(np.int64))).numpy().astype(np.uint32), words)
    assert np.array_equal(rs.get_state()[1], nxt) and rs.get_state()[2] == 624


@pytest.mark.parametrize("seed,burn,requests", [
    (0, 0, [(1000, 25.5)]),
    (1, 7, [(999, 10.2), (1, 3.0), (2, 1.0), (1501, 12.75)]),             # odd counts: the cached normal travels between draws
    (2, 623, [(5, 1.0), (0, 1.0), (4, 2.0)]),                              # starts on the last word of a block
    (3, 1, [(37 * 53 * 3, 0.05 * 255), (37 * 53 * 3, 0.1 * 255)]),
    (4, 11, [(1, 5.0), (1, 5.0), (1, 5.0), (3, 2.5)]),
])
def test_device_evaluation_equals_np_random_normal(seed, burn, requests):
    np.random.seed(seed)
    if burn:
        np.random.random_sample(burn)                                      # consumes 2 words each
    if seed == 1:
        np.random.normal(0, 1, 3)                                          # leaves a cached normal behind
    st = np.random.get_state()
    _, key, pos, has_gauss, gauss = st
    want = [np.random.normal(0, s, n).astype(np.float32) for n, s in requests]
    after = np.random.get_state()
    total = NS.words_needed(sum(n for n, _ in requests)) + 700
    raw = raw_stream(key, total // 624 + 2)
    d = NS.normals(raw, pos, bool(has_gauss), float(gauss), requests)
    for got, w in zip(d.noise, want):
        assert np.array_equal(got.numpy(), w)
    k, p = NS.state_at(raw, d.position, pos)
    assert p == after[2] and np.array_equal(k, after[1])
    assert int(d.has_gauss) == after[3] and (float(d.gauss) == after[4] or not d.has_gauss)


def test_samples_near_a_float32_rounding_boundary_are_recomputed_on_the_host(monkeypatch):
    """With the margin blown up to 2^-26 a quarter of the samples take the host path: the results must not change."""
    np.random.seed(21)
    _, key, pos, has_gauss, gauss = np.random.get_state()
    want = np.random.normal(0, 12.75, 5001).astype(np.float32)
    raw = raw_stream(key, NS.words_needed(5001) // 624 + 3)
    monkeypatch.setattr(NS, "MARGIN", 2.0 ** -26)
    d = NS.normals(raw, pos, bool(has_gauss), float(gauss), [(5001, 12.75)])
    assert d.patched > 500 and np.array_equal(d.noise[0].numpy(), want)
