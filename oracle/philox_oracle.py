"""Philox4x32-10 in NumPy (TEST INFRASTRUCTURE ONLY): the counter-based generator behind the opt-in device noise
(imgxf_add_noise_philox_u8, csrc/noise_rng.hip).  Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as
1, 2, 3" (SC'11); constants and the known-answer vectors are Random123's (kat_vectors: philox4x32 10 ...)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(counter, key):
    """counter: [..., 4] uint32, key: [..., 2] uint32 -> [..., 4] uint32."""
    c = np.array(counter, dtype=np.uint32, copy=True)
    k = np.array(key, dtype=np.uint32, copy=True)
    c0, c1, c2, c3 = (c[..., i].copy() for i in range(4))
    k0, k1 = k[..., 0].copy(), k[..., 1].copy()
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = p1.astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = p0.astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = k0 + W0
            k1 = k1 + W1
    return np.stack([c0, c1, c2, c3], axis=-1)


def stream(count, seed, offset=0):
    """The uint32 stream of imgxf_philox4x32_u32: block t uses counter (lo, hi, 0, 0) of t + offset / 4, key = seed."""
    assert count % 4 == 0 and offset % 4 == 0
    t = np.arange(count // 4, dtype=np.uint64) + np.uint64(offset // 4)
    ctr = np.zeros((count // 4, 4), np.uint32)
    ctr[:, 0] = (t & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[:, 1] = (t >> np.uint64(32)).astype(np.uint32)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], np.uint32)
    return philox4x32_10(ctr, np.broadcast_to(key, (count // 4, 2))).reshape(-1)


KAT = [  # Random123 kat_vectors, philox4x32 10 rounds: (counter, key, expected)
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]
