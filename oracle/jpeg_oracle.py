"""TEST INFRASTRUCTURE ONLY — CPU restatement of the JPEG writer behind the reference driver's
`transformed.save(path)` (`transformation.py:161-162`).

The algorithm lives in third-party dependencies absent from /root/reference: Pillow (`JpegImagePlugin._save`, defaults:
quality 75, 4:2:0, no optimisation, no progression) over libjpeg-turbo (this image: Pillow 12.2.0 / libjpeg-turbo, 6.2
API).  Restated from the library's published algorithm: `jccolor.c` rgb_ycc_convert (16-bit fixed point), `jcsample.c`
h2v2_downsample / fullsize_downsample with `expand_right_edge`, `jcprepct.c` pre_process_data (bottom padding of the
DOWNSAMPLED rows), `jfdctint.c` jpeg_fdct_islow, `jcdctmgr.c` quantize (round half away from zero of coefficient /
(8·q)), `jccoefct.c` compress_data (dummy blocks: zero AC, DC of the previous block), `jchuff.c` encode_one_block with
the Annex-K tables, `jcmarker.c` marker order.  Pinned: `tests/test_jpeg_oracle.py` compares the byte stream with
Pillow's own output on seeded images (every size class: MCU-aligned, odd, 1-pixel, dummy-block cases) and with the
sha256 fixtures in `tests/golden/jpeg_q75.json`.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module."""
import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,
                   7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                   39, 46, 53, 60, 61, 54, 47, 55, 62, 63])
STD_LUM_Q = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
                      14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                      49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99])
STD_CHR_Q = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
                      47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32)
DC_LUM_BITS = [0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]
DC_CHR_BITS = [0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0]
DC_VALS = list(range(12))
AC_LUM_BITS = [0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 125]
AC_LUM_VALS = [
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32,
    0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16,
    0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45,
    0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
    0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94,
    0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
    0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8,
    0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
    0xf9, 0xfa]
AC_CHR_BITS = [0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 119]
AC_CHR_VALS = [
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81,
    0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34,
    0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44,
    0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68,
    0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92,
    0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
    0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
    0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
    0xf9, 0xfa]


def quant_tables(quality=75):
    """jcparam.c jpeg_set_quality → jpeg_quality_scaling + jpeg_add_quant_table(force_baseline); natural order."""
    q = min(max(int(quality), 1), 100)
    scale = 5000 // q if q < 50 else 200 - 2 * q
    out = []
    for base in (STD_LUM_Q, STD_CHR_Q):
        t = (base.astype(np.int64) * scale + 50) // 100
        out.append(np.clip(t, 1, 255).astype(np.int64))
    return out


def huff_codes(bits, vals):
    """jchuff.c jpeg_make_c_derived_tbl: canonical codes in order of increasing length; returns {symbol: (code, len)}."""
    table, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(bits[length - 1]):
            table[vals[k]] = (code, length)
            code += 1
            k += 1
        code <<= 1
    return table


def header(w, h, qt):
    """jcmarker.c: SOI, APP0 (JFIF 1.01, density 1:1 unit 0), DQT×2 (8-bit, zigzag), SOF0 (2x2,1x1,1x1), DHT×4, SOS."""
    out = bytearray(b"\xff\xd8\xff\xe0\x00\x10JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00")
    for i, t in enumerate(qt):
        out += b"\xff\xdb\x00\x43" + bytes([i]) + bytes(int(t[z]) for z in ZIGZAG)
    out += b"\xff\xc0\x00\x11\x08" + h.to_bytes(2, "big") + w.to_bytes(2, "big") + b"\x03\x01\x22\x00\x02\x11\x01\x03\x11\x01"
    for cls, bits, vals in ((0x00, DC_LUM_BITS, DC_VALS), (0x10, AC_LUM_BITS, AC_LUM_VALS),
                            (0x01, DC_CHR_BITS, DC_VALS), (0x11, AC_CHR_BITS, AC_CHR_VALS)):
        out += b"\xff\xc4" + (19 + len(vals)).to_bytes(2, "big") + bytes([cls]) + bytes(bits) + bytes(vals)
    out += b"\xff\xda\x00\x0c\x03\x01\x00\x02\x11\x03\x11\x00\x3f\x00"
    return bytes(out)


def ycc_planes(img):
    """jccolor.c rgb_ycc_convert: FIX(x) = int(x·65536 + 0.5); Cb / Cr carry 128<<16 and ONE_HALF − 1."""
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    y = (19595 * r + 38470 * g + 7471 * b + 32768) >> 16
    cb = (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16
    cr = (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16
    return y, cb, cr


def padded_luma(y, bw, bh):
    """fullsize_downsample + expand_right_edge, expand_bottom_edge: replicate the last column / row."""
    h, w = y.shape
    return y[np.minimum(np.arange(bh * 8), h - 1)][:, np.minimum(np.arange(bw * 8), w - 1)]


def padded_chroma(c, bw, bh):
    """h2v2_downsample: the INPUT's last column is replicated to 16·bw columns, bias 1,2,1,2 along a row, >> 2; rows
    beyond the image: pre_process_data first fills the row pair with the last input row, then replicates the last
    DOWNSAMPLED row down the iMCU row."""
    h, w = c.shape
    rows = (h + 1) // 2
    r0 = np.minimum(2 * np.arange(rows), h - 1)
    r1 = np.minimum(2 * np.arange(rows) + 1, h - 1)
    cols = np.minimum(np.arange(bw * 16), w - 1)
    a, b = c[r0][:, cols], c[r1][:, cols]
    bias = np.tile(np.array([1, 2]), bw * 4)
    ds = (a[:, 0::2] + a[:, 1::2] + b[:, 0::2] + b[:, 1::2] + bias) >> 2
    return ds[np.minimum(np.arange(bh * 8), rows - 1)]


def fdct_islow(blocks):
    """jfdctint.c jpeg_fdct_islow on [..., 8, 8] int64 samples already centred (− 128); output scaled by 8."""
    C = dict(f0298=2446, f0390=3196, f0541=4433, f0765=6270, f0899=7373, f1175=9633, f1501=12299, f1847=15137,
             f1961=16069, f2053=16819, f2562=20995, f3072=25172)

    def descale(x, n):
        return (x + (1 << (n - 1))) >> n

    def one_d(d, first):
        t0, t7 = d[0] + d[7], d[0] - d[7]
        t1, t6 = d[1] + d[6], d[1] - d[6]
        t2, t5 = d[2] + d[5], d[2] - d[5]
        t3, t4 = d[3] + d[4], d[3] - d[4]
        t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
        o = [None] * 8
        if first:
            o[0], o[4] = (t10 + t11) << 2, (t10 - t11) << 2
            n = 11
        else:
            o[0], o[4] = descale(t10 + t11, 2), descale(t10 - t11, 2)
            n = 15
        z1 = (t12 + t13) * C["f0541"]
        o[2] = descale(z1 + t13 * C["f0765"], n)
        o[6] = descale(z1 - t12 * C["f1847"], n)
        z1, z2, z3, z4 = t4 + t7, t5 + t6, t4 + t6, t5 + t7
        z5 = (z3 + z4) * C["f1175"]
        t4, t5, t6, t7 = t4 * C["f0298"], t5 * C["f2053"], t6 * C["f3072"], t7 * C["f1501"]
        z1, z2, z3, z4 = -z1 * C["f0899"], -z2 * C["f2562"], -z3 * C["f1961"] + z5, -z4 * C["f0390"] + z5
        o[7], o[5], o[3], o[1] = descale(t4 + z1 + z3, n), descale(t5 + z2 + z4, n), descale(t6 + z2 + z3, n), descale(t7 + z1 + z4, n)
        return o

    rows = one_d([blocks[..., :, i] for i in range(8)], True)          # pass 1: along each row
    x = np.stack(rows, axis=-1)
    cols = one_d([x[..., i, :] for i in range(8)], False)              # pass 2: along each column
    return np.stack(cols, axis=-2)


def quantise(coef, q):
    """jcdctmgr.c quantize: sign · ((|c| + 4q) // 8q) (islow output carries a factor 8)."""
    d = (q.reshape(8, 8) * 8).astype(np.int64)
    a = np.abs(coef)
    return np.sign(coef) * ((a + (d >> 1)) // d)


def coefficients(img, quality=75):
    """Quantised coefficients of the REAL blocks, natural order: (Y [bh, bw, 64], Cb, Cr [ch, cw, 64])."""
    h, w, _ = img.shape
    qt = quant_tables(quality)
    y, cb, cr = ycc_planes(img)
    bw, bh = (w + 7) // 8, (h + 7) // 8
    cw, ch = ((w + 1) // 2 + 7) // 8, ((h + 1) // 2 + 7) // 8
    out = []
    for plane, nbw, nbh, q in ((padded_luma(y, bw, bh), bw, bh, qt[0]), (padded_chroma(cb, cw, ch), cw, ch, qt[1]),
                               (padded_chroma(cr, cw, ch), cw, ch, qt[1])):
        blocks = plane.reshape(nbh, 8, nbw, 8).transpose(0, 2, 1, 3) - 128
        out.append(quantise(fdct_islow(blocks), q).reshape(nbh, nbw, 64))
    return out


class _Bits:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, code, length):
        self.acc = (self.acc << length) | (code & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            byte = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(byte)
            if byte == 0xFF:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put(0x7F, 8 - self.n)


def mcu_blocks(img, quality=75):
    """jccoefct.c compress_data: per MCU the six blocks (zigzag order) incl. dummy blocks; yields (component, block)."""
    h, w, _ = img.shape
    Y, Cb, Cr = coefficients(img, quality)
    bh, bw = Y.shape[:2]
    mw, mh = (w + 15) // 16, (h + 15) // 16
    for my in range(mh):
        for mx in range(mw):
            prev = None
            for yi in range(2):
                for xi in range(2):
                    by, bx = 2 * my + yi, 2 * mx + xi
                    if by < bh and bx < bw:
                        blk = Y[by, bx][ZIGZAG]
                    else:                                    # dummy: zero AC, DC of the block before it in the MCU
                        blk = np.zeros(64, np.int64)
                        blk[0] = prev[0]
                    prev = blk
                    yield 0, blk
            yield 1, Cb[my, mx][ZIGZAG]
            yield 2, Cr[my, mx][ZIGZAG]


def encode(img, quality=75):
    """The whole file as Pillow's `Image.save(fp, "JPEG")` writes it for an RGB image."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, _ = img.shape
    dc = [huff_codes(DC_LUM_BITS, DC_VALS), huff_codes(DC_CHR_BITS, DC_VALS)]
    ac = [huff_codes(AC_LUM_BITS, AC_LUM_VALS), huff_codes(AC_CHR_BITS, AC_CHR_VALS)]
    bits, last = _Bits(), [0, 0, 0]
    for comp, blk in mcu_blocks(img, quality):
        t = 0 if comp == 0 else 1
        diff = int(blk[0]) - last[comp]
        last[comp] = int(blk[0])
        mag = abs(diff).bit_length()
        bits.put(*dc[t][mag])
        if mag:
            bits.put(diff if diff >= 0 else diff - 1, mag)
        run = 0
        for k in range(1, 64):
            v = int(blk[k])
            if v == 0:
                run += 1
                continue
            while run > 15:
                bits.put(*ac[t][0xF0])
                run -= 16
            mag = abs(v).bit_length()
            bits.put(*ac[t][(run << 4) | mag])
            bits.put(v if v >= 0 else v - 1, mag)
            run = 0
        if run:
            bits.put(*ac[t][0x00])
    bits.flush()
    return header(w, h, quant_tables(quality)) + bytes(bits.out) + b"\xff\xd9"
