"""The jump polynomial of MT19937 for a stride of J words (development tool; writes imagetransformations_amd/mt19937_jump.npz).

F = the one-word step of the generator on its canonical state (S[t], ..., S[t+623]) -> (S[t+1], ..., S[t+624]),
linear over GF(2) on 19937 bits.  phi = its characteristic polynomial, found with Berlekamp-Massey on one output bit of the
raw word stream; g(x) = x^J mod phi.  Then F^J s = g(F) s = sum_i g_i F^i s (Haramoto, Matsumoto, Nishimura, Panneton,
L'Ecuyer: "Efficient jump ahead for F2-linear random number generators"), evaluated by Horner's rule on the device
(imgxf_mt19937_jump).  Checked here against J sequential steps in NumPy before the file is written.
    python tools/make_mt_jump.py [log2 of the number of 624-word blocks per stride, default 16]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagetransformations_amd.numpy_stream import mt_next_block

K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
J = 624 * (1 << K)
DEG = 19937


def bit_sequence(n):
    """bit 0 of the raw state words S[0 .. n)"""
    key = np.random.RandomState(4357).get_state()[1].astype(np.uint32)
    out, blk = [], key
    while len(out) * 624 < n:
        out.append(blk & 1)
        blk = mt_next_block(blk)
    return np.concatenate(out)[:n].astype(np.uint8)


def berlekamp_massey(bits):
    """minimal polynomial (as an int, bit i = coefficient of x^i, forward shift: sum_i phi_i s[u + i] = 0) of the binary
    sequence and its degree"""
    C, B, L, m = 1, 1, 0, 1                      # connection polynomial: sum_{i=0..L} c_i s[t - i] = 0, c_0 = 1
    R = 0                                        # bit j = s[t - j]: the newest element at bit 0
    for t in range(len(bits)):
        R = (R << 1) | int(bits[t])
        if bin(C & R).count("1") & 1:            # discrepancy
            T = C
            C ^= B << m
            if 2 * L <= t:
                L, B, m = t + 1 - L, T, 1
            else:
                m += 1
        else:
            m += 1
    phi = int(format(C, "0%db" % (L + 1))[::-1], 2)      # c_i -> coefficient of x^(L - i)
    return phi, L


def poly_mod(a, phi, deg):
    while a.bit_length() - 1 >= deg:
        a ^= phi << (a.bit_length() - 1 - deg)
    return a


_SPREAD = [int("".join(c + "0" for c in format(b, "08b"))[:-1] if False else format(b, "08b").replace("0", "00").replace("1", "01"), 2) for b in range(256)]


def poly_square(a):
    """a(x)^2 over GF(2): bit i -> bit 2 i"""
    out, shift = 0, 0
    while a:
        out |= _SPREAD[a & 0xFF] << shift
        a >>= 8
        shift += 16
    return out


def main():
    t0 = time.time()
    bits = bit_sequence(2 * DEG + 201)[1:]       # (bit 0 of S[0] is not part of the 19937-bit state: only its top bit is)
    phi, L = berlekamp_massey(bits)
    print("degree of the minimal polynomial:", L, f"({time.time() - t0:.1f} s)")
    assert L == DEG
    # check: the recurrence holds on the tail of the sequence
    idx = [i for i in range(DEG + 1) if (phi >> i) & 1]
    for t in (0, 17, 150):
        assert sum(int(bits[t + i]) for i in idx) % 2 == 0
    # g_m = x^(m J) mod phi for m = 1 .. M: g_1 by square and multiply, g_(m+1) = g_m g_1 mod phi
    def poly_mul(a, b):
        out, sh = 0, 0
        while b:
            if b & 1:
                out ^= a << sh
            b >>= 1; sh += 1
        return out
    g1 = 1
    for b in bin(J)[2:]:
        g1 = poly_mod(poly_square(g1), phi, DEG)
        if b == "1":
            g1 = poly_mod(g1 << 1, phi, DEG)
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    gs = [g1]
    for m in range(2, M + 1):
        gs.append(poly_mod(poly_mul(gs[-1], g1), phi, DEG))
    print(f"jump polynomials for m J, J = {J} words, m = 1 .. {M}: {[bin(g).count('1') for g in gs][:4]} ... terms ({time.time() - t0:.1f} s)")
    # check by Horner on a random key against sequential blocks (m = 1 and m = 3)
    key = np.random.RandomState(99).get_state()[1].astype(np.uint32)
    s = key.astype(np.uint64)
    blk, want = key, {}
    for b in range(1, 3 * (1 << K) + 1):
        blk = mt_next_block(blk)
        if b % (1 << K) == 0:
            want[b >> K] = blk.astype(np.uint64)
    for m in (1, 3):
        g = gs[m - 1]
        r = np.zeros(624, np.uint64)
        for i in range(DEG - 1, -1, -1):
            y = (r[0] & 0x80000000) | (r[1] & 0x7FFFFFFF)
            new = r[397] ^ (y >> np.uint64(1)) ^ (np.uint64(0x9908B0DF) if (int(y) & 1) else np.uint64(0))
            r[:-1] = r[1:]; r[-1] = new
            if (g >> i) & 1:
                r ^= s
        ok = np.array_equal(r[1:] & 0xFFFFFFFF, want[m][1:]) and (int(r[0]) & 0x80000000) == (int(want[m][0]) & 0x80000000)
        print(f"Horner on the host, m = {m}: equals {m << K} sequential blocks:", ok, f"({time.time() - t0:.1f} s)")
        assert ok
    coef = np.zeros((M, 2496), np.uint8)
    for m, g in enumerate(gs):
        bits = np.array([(g >> i) & 1 for i in range(DEG)], np.uint8)
        packed = np.packbits(bits, bitorder="little")
        coef[m, :len(packed)] = packed
    np.savez_compressed(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "imagetransformations_amd", "mt19937_jump.npz"),
                        log2_blocks=np.int64(K), coef=coef)
    print("written")


if __name__ == "__main__":
    main()
