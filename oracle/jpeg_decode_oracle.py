"""TEST INFRASTRUCTURE ONLY — CPU restatement of the JPEG READER behind the reference's load step
`Image.open(path).convert("RGB")` (/root/reference/transformation.py:83; fall_2025/TTA_transforms.py:16-36; SURVEY 8f
row 4, decode half).

The algorithm lives in third-party dependencies absent from /root/reference: Pillow (JpegImagePlugin, no draft mode,
no scaling) over libjpeg-turbo (this image: Pillow 12.2.0 / libjpeg-turbo, 6.2 API) with its defaults: JDCT_ISLOW,
do_fancy_upsampling = TRUE, no colour quantisation.  Restated from the library's published algorithm:
`jdmarker.c` (SOI / APPn / DQT / SOF0-1 / DHT / DRI / SOS / EOI), `jdhuff.c` decode_mcu (DC differences per component,
AC run / size, EOB, ZRL; byte stuffing removed, RSTn resynchronisation), `jdcoefct.c` (interleaved MCU order, dummy
blocks at the right / bottom edge are decoded and dropped), `jidctint.c` jpeg_idct_islow (13-bit constants, 2 pass
bits, the masked range-limit table), `jdsample.c` fullsize / h2v1_fancy / h2v2_fancy upsampling with the edge
replication of `jdmainct.c`, `jdcolor.c` ycc_rgb_convert (16-bit fixed-point tables) and grayscale -> RGB replication
(Pillow's convert("RGB") of mode "L").  Pinned: tests/test_jpeg_decode_oracle.py compares the pixels with Pillow's own
decoder on the 30 files the reference itself wrote (tests/golden/reference_outputs/) and on seeded images of every
size class, sampling (4:4:4, 4:2:2, 4:2:0, grayscale), quality and table kind (standard / optimised, restart intervals).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module."""
import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,
                   7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                   39, 46, 53, 60, 61, 54, 47, 55, 62, 63])


class Unsupported(ValueError):
    """A JPEG this restatement (and the device reader) does not cover: progressive, arithmetic, 12-bit, CMYK, ..."""


def parse(data: bytes):
    """Marker segments up to the first scan -> dict(width, height, comps=[(id, h, v, tq)], qt={id: [64] natural order},
    huff={(class, id): (bits[16], vals)}, scan=[(comp index, td, ta)], dri, ecs=(start, end) of the entropy-coded bytes)."""
    if data[:2] != b"\xff\xd8":
        raise Unsupported("not a JPEG (no SOI)")
    pos, qt, huff, frame, dri = 2, {}, {}, None, 0
    while True:
        if pos + 4 > len(data):
            raise Unsupported("truncated before SOS")
        if data[pos] != 0xFF:
            raise Unsupported("marker expected")
        while data[pos + 1] == 0xFF:                    # fill bytes
            pos += 1
        marker = data[pos + 1]
        seglen = int.from_bytes(data[pos + 2:pos + 4], "big")
        seg = data[pos + 4:pos + 2 + seglen]
        if marker == 0xDB:                              # DQT
            i = 0
            while i < len(seg):
                pq, tq = seg[i] >> 4, seg[i] & 15
                i += 1
                if pq:
                    vals = [int.from_bytes(seg[i + 2 * k:i + 2 * k + 2], "big") for k in range(64)]
                    i += 128
                else:
                    vals = list(seg[i:i + 64])
                    i += 64
                t = np.zeros(64, np.int64)
                t[ZIGZAG] = vals                        # the file holds zigzag order
                qt[tq] = t
        elif marker in (0xC0, 0xC1):                    # SOF0 baseline / SOF1 extended sequential (Huffman)
            if seg[0] != 8:
                raise Unsupported("sample precision %d" % seg[0])
            h, w, nc = int.from_bytes(seg[1:3], "big"), int.from_bytes(seg[3:5], "big"), seg[5]
            comps = [(seg[6 + 3 * k], seg[7 + 3 * k] >> 4, seg[7 + 3 * k] & 15, seg[8 + 3 * k]) for k in range(nc)]
            frame = (w, h, comps)
        elif marker in (0xC2, 0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise Unsupported("SOF%d (progressive / lossless / arithmetic)" % (marker - 0xC0))
        elif marker == 0xC4:                            # DHT
            i = 0
            while i < len(seg):
                tc, th = seg[i] >> 4, seg[i] & 15
                bits = list(seg[i + 1:i + 17])
                n = sum(bits)
                huff[(tc, th)] = (bits, list(seg[i + 17:i + 17 + n]))
                i += 17 + n
        elif marker == 0xDD:
            dri = int.from_bytes(seg[0:2], "big")
        elif marker == 0xDA:                            # SOS
            if frame is None:
                raise Unsupported("SOS before SOF")
            ns = seg[0]
            scan = []
            for k in range(ns):
                cid, tt = seg[1 + 2 * k], seg[2 + 2 * k]
                idx = [c[0] for c in frame[2]].index(cid)
                scan.append((idx, tt >> 4, tt & 15))
            if ns != len(frame[2]):
                raise Unsupported("non-interleaved scans")
            start = pos + 2 + seglen
            end = start
            while True:                                 # the scan ends at the first marker that is not RSTn / stuffing
                end = data.index(b"\xff", end)
                nxt = data[end + 1]
                if nxt == 0x00 or 0xD0 <= nxt <= 0xD7:
                    end += 2
                    continue
                break
            w, h, comps = frame
            if len(comps) not in (1, 3):
                raise Unsupported("%d components" % len(comps))
            return dict(width=w, height=h, comps=comps, qt=qt, huff=huff, scan=scan, dri=dri, ecs=(start, end))
        pos += 2 + seglen


def _derive(bits, vals):
    """jdhuff.c jpeg_make_d_derived_tbl: canonical codes -> {(length, code): symbol}."""
    table, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(bits[length - 1]):
            table[(length, code)] = vals[k]
            code += 1
            k += 1
        code <<= 1
    return table


def decode_coefficients(data: bytes, info=None):
    """Entropy decoding: -> (info, [per component int16 [blocks_y, blocks_x, 64] in NATURAL order, not dequantised])."""
    info = info or parse(data)
    w, h, comps = info["width"], info["height"], info["comps"]
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    mcux, mcuy = -(-w // (8 * hmax)), -(-h // (8 * vmax))
    if len(comps) == 1:                                 # a single-component scan is never interleaved: MCU = one block
        hmax = vmax = 1
        comps = [(comps[0][0], 1, 1, comps[0][3])]
        mcux, mcuy = -(-w // 8), -(-h // 8)
    coefs = [np.zeros((mcuy * c[2], mcux * c[1], 64), np.int16) for c in comps]
    tabs = {k: _derive(*v) for k, v in info["huff"].items()}
    raw = data[info["ecs"][0]:info["ecs"][1]]
    # split at RSTn markers, then remove the stuffing
    segs, cur, i = [], bytearray(), 0
    while i < len(raw):
        b = raw[i]
        if b == 0xFF and i + 1 < len(raw):
            n = raw[i + 1]
            if n == 0:
                cur.append(0xFF); i += 2; continue
            if 0xD0 <= n <= 0xD7:
                segs.append(bytes(cur)); cur = bytearray(); i += 2; continue
        cur.append(b); i += 1
    segs.append(bytes(cur))
    dri = info["dri"] or (mcux * mcuy)
    mcu = 0
    for seg in segs:
        acc, accbits, bpos = 0, 0, 0                  # bit reader: a small accumulator refilled a byte at a time; past
                                                        # the end of the segment it reads zero bits (jdhuff.c pads likewise)
        def take(n):
            nonlocal acc, accbits, bpos
            while accbits < n:
                acc = (acc << 8) | (seg[bpos] if bpos < len(seg) else 0)
                bpos += 1
                accbits += 8
            accbits -= n
            v = (acc >> accbits) & ((1 << n) - 1)
            acc &= (1 << accbits) - 1
            return v

        def symbol(tab):
            code = 0
            for length in range(1, 17):
                code = (code << 1) | take(1)
                s = tab.get((length, code))
                if s is not None:
                    return s
            raise Unsupported("bad Huffman code")

        pred = [0] * len(comps)
        for _ in range(dri):
            if mcu >= mcux * mcuy:
                break
            my, mx = divmod(mcu, mcux)
            for ci, (idx, td, ta) in enumerate(info["scan"]):
                _, ch, cv, _ = comps[idx]
                for by in range(cv):
                    for bx in range(ch):
                        blk = coefs[idx][my * cv + by, mx * ch + bx]
                        s = symbol(tabs[(0, td)])
                        diff = take(s)
                        if s and diff < (1 << (s - 1)):
                            diff -= (1 << s) - 1
                        pred[idx] += diff
                        blk[0] = np.int16(pred[idx])
                        k = 1
                        while k < 64:
                            rs = symbol(tabs[(1, ta)])
                            r, s = rs >> 4, rs & 15
                            if s == 0:
                                if r == 15:
                                    k += 16
                                    continue
                                break
                            k += r
                            v = take(s)
                            if v < (1 << (s - 1)):
                                v -= (1 << s) - 1
                            blk[ZIGZAG[k & 63]] = np.int16(v)       # (k & 63: a corrupt run cannot leave the block)
                            k += 1
            mcu += 1
    return info, coefs


# ---- jidctint.c jpeg_idct_islow ---------------------------------------------------------------------------------
C_BITS, P1 = 13, 2
F = dict(a=2446, b=3196, c=4433, d=6270, e=7373, f=9633, g=12299, h=15137, i=16069, j=16819, k=20995, m=25172)
# FIX_0_298631336=2446 0_390180644=3196 0_541196100=4433 0_765366865=6270 0_899976223=7373 1_175875602=9633
# 1_501321110=12299 1_847759065=15137 1_961570560=16069 2_053119869=16819 2_562915447=20995 3_072711026=25172


def _idct_1d(x0, x1, x2, x3, x4, x5, x6, x7, shift, pass1):
    z2, z3 = x2, x6
    z1 = (z2 + z3) * F["c"]
    tmp2 = z1 + z3 * (-F["h"])
    tmp3 = z1 + z2 * F["d"]
    z2, z3 = x0, x4
    tmp0 = (z2 + z3) << C_BITS
    tmp1 = (z2 - z3) << C_BITS
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = x7, x5, x3, x1
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * F["f"]
    tmp0, tmp1, tmp2, tmp3 = tmp0 * F["a"], tmp1 * F["j"], tmp2 * F["m"], tmp3 * F["g"]
    z1, z2, z3, z4 = z1 * (-F["e"]), z2 * (-F["k"]), z3 * (-F["i"]) + z5, z4 * (-F["b"]) + z5
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4
    rnd = 1 << (shift - 1)
    outs = [tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3]
    return [(o + rnd) >> shift for o in outs]


def range_limit(x):
    """sample_range_limit + CENTERJSAMPLE indexed with (x & RANGE_MASK): jdmaster.c prepare_range_limit_table."""
    idx = x & 1023
    return np.where(idx < 128, idx + 128, np.where(idx < 512, 255, np.where(idx < 896, 0, idx - 896))).astype(np.uint8)


def idct_islow(coef, quant):
    """coef [..., 64] int16 natural order, quant [64] -> [..., 8, 8] uint8 samples."""
    x = coef.astype(np.int64) * quant.astype(np.int64)
    x = x.reshape(x.shape[:-1] + (8, 8))                # [row, col]
    cols = _idct_1d(*[x[..., r, :] for r in range(8)], C_BITS - P1, True)          # pass 1: columns -> ws[row][col]
    ws = np.stack(cols, axis=-2)
    rows = _idct_1d(*[ws[..., :, c] for c in range(8)], C_BITS + P1 + 3, False)   # pass 2: rows
    out = np.stack(rows, axis=-1)
    return range_limit(out)


def _plane(blocks):
    by, bx = blocks.shape[:2]
    return blocks.transpose(0, 2, 1, 3).reshape(by * 8, bx * 8)


def _h2v1_fancy(p, dw):
    """jdsample.c h2v1_fancy_upsample on rows of a plane whose real width is dw (> 2) -> 2 dw columns."""
    p = p[:, :dw].astype(np.int64)
    out = np.empty((p.shape[0], 2 * dw), np.int64)
    left = np.concatenate([p[:, :1], p[:, :-1]], axis=1)
    right = np.concatenate([p[:, 1:], p[:, -1:]], axis=1)
    out[:, 0::2] = (3 * p + left + 1) >> 2
    out[:, 1::2] = (3 * p + right + 2) >> 2
    out[:, 0] = p[:, 0]
    out[:, -1] = p[:, -1]
    return out


def _h2v2_fancy(p, dw, dh):
    """jdsample.c h2v2_fancy_upsample with the context rows of jdmainct.c (edge rows replicated): plane with real size
    dh x dw -> 2 dh x 2 dw."""
    p = p[:dh, :dw].astype(np.int64)
    above = np.concatenate([p[:1], p[:-1]], axis=0)
    below = np.concatenate([p[1:], p[-1:]], axis=0)
    out = np.empty((2 * dh, 2 * dw), np.int64)
    for v, near in ((0, above), (1, below)):
        colsum = 3 * p + near                           # thiscolsum for every column
        last = np.concatenate([colsum[:, :1], colsum[:, :-1]], axis=1)
        nxt = np.concatenate([colsum[:, 1:], colsum[:, -1:]], axis=1)
        even = (3 * colsum + last + 8) >> 4
        odd = (3 * colsum + nxt + 7) >> 4
        even[:, 0] = (colsum[:, 0] * 4 + 8) >> 4
        odd[:, -1] = (colsum[:, -1] * 4 + 7) >> 4
        out[v::2, 0::2] = even
        out[v::2, 1::2] = odd
    return out


def _ycc_tables():
    x = np.arange(256, dtype=np.int64) - 128
    fix = lambda v: int(v * 65536 + 0.5)
    half = 1 << 15
    return ((fix(1.40200) * x + half) >> 16, (fix(1.77200) * x + half) >> 16, -fix(0.71414) * x, -fix(0.34414) * x + half)


def decode(data: bytes) -> np.ndarray:
    """The pixels of Image.open(BytesIO(data)).convert("RGB") as an [H, W, 3] uint8 array."""
    info, coefs = decode_coefficients(data)
    w, h, comps = info["width"], info["height"], info["comps"]
    planes = [_plane(idct_islow(coefs[i], info["qt"][comps[i][3]])) for i in range(len(comps))]
    if len(comps) == 1:
        y = planes[0][:h, :w]
        return np.stack([y, y, y], axis=-1)
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    full = []
    for (cid, ch, cv, tq), p in zip(comps, planes):
        dw, dh = -(-w * ch // hmax), -(-h * cv // vmax)   # downsampled_width / height of the component
        if ch == hmax and cv == vmax:
            up = p.astype(np.int64)
        elif ch * 2 == hmax and cv == vmax:
            # jdsample.c jinit_upsampler: the fancy (triangle) filters are chosen only when downsampled_width > 2,
            # narrower components are replicated (h2v1_upsample / h2v2_upsample)
            up = _h2v1_fancy(p[:dh], dw) if dw > 2 else np.repeat(p[:dh, :dw].astype(np.int64), 2, axis=1)
        elif ch * 2 == hmax and cv * 2 == vmax:
            up = _h2v2_fancy(p, dw, dh) if dw > 2 else np.repeat(np.repeat(p[:dh, :dw].astype(np.int64), 2, axis=0), 2, axis=1)
        else:
            raise Unsupported("sampling %dx%d of %dx%d" % (ch, cv, hmax, vmax))
        full.append(up[:h, :w])
    y, cb, cr = (f.astype(np.int64) for f in full)
    cr_r, cb_b, cr_g, cb_g = _ycc_tables()
    clamp = lambda v: np.clip(v, 0, 255).astype(np.uint8)   # range_limit[y + x] with y in 0..255 and |x| < 256 is a plain clamp
    r = clamp(y + cr_r[cr])
    g = clamp(y + ((cb_g[cb] + cr_g[cr]) >> 16))
    b = clamp(y + cb_b[cb])
    return np.stack([r, g, b], axis=-1)
