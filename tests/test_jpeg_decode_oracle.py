"""CPU: the JPEG reader restatement (oracle/jpeg_decode_oracle.py) against Pillow / libjpeg-turbo, the codec behind
the reference's `Image.open(path).convert("RGB")` (/root/reference/transformation.py:83): bit-identical pixels on
files the reference itself wrote (tests/golden/reference_outputs/*.JPEG) and on seeded images of every size class,
chroma sampling, quality and table kind."""
import glob
import io
import os

import numpy as np
import pytest
from PIL import Image

from conftest import synth
from oracle import jpeg_decode_oracle as JD

REF = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "reference_outputs", "*.JPEG")))


def pillow_rgb(data: bytes) -> np.ndarray:
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


def photo_like(seed, h, w):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 128 + 70 * np.sin(xx / 9.0 + seed) + 50 * np.cos(yy / 7.0)
    img = base[..., None] + rng.normal(0, 12, (h, w, 3)) + np.array([10, -20, 30])
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("path", REF[::5], ids=lambda p: os.path.basename(p)[:40])
def test_reference_written_files(path):
    data = open(path, "rb").read()
    assert np.array_equal(JD.decode(data), pillow_rgb(data))


@pytest.mark.parametrize("h,w", [(1, 1), (7, 5), (8, 8), (16, 16), (17, 33), (31, 15), (48, 64), (100, 75),
                                 (9, 2), (40, 3), (174, 4), (21, 6), (2, 40)])    # widths <= 4: chroma narrower than 3 samples is replicated
@pytest.mark.parametrize("subsampling", [0, 1, 2])                      # 4:4:4, 4:2:2, 4:2:0
def test_sizes_and_samplings(h, w, subsampling):
    for seed, quality in ((1, 75), (2, 30), (3, 95)):
        for img in (synth(seed * 7 + h, h, w), photo_like(seed, h, w)):
            buf = io.BytesIO()
            Image.fromarray(img).save(buf, "JPEG", quality=quality, subsampling=subsampling)
            data = buf.getvalue()
            assert np.array_equal(JD.decode(data), pillow_rgb(data)), (h, w, subsampling, quality)


def test_grayscale_optimised_tables_and_restart_intervals():
    img = photo_like(5, 61, 83)
    for kwargs in (dict(optimize=True), dict(quality=10), dict(quality=100, subsampling=0), dict(restart_marker_blocks=3),
                   dict(restart_marker_rows=1, subsampling=2)):
        buf = io.BytesIO()
        try:
            Image.fromarray(img).save(buf, "JPEG", **kwargs)
        except TypeError:
            continue
        data = buf.getvalue()
        assert np.array_equal(JD.decode(data), pillow_rgb(data)), kwargs
    buf = io.BytesIO()
    Image.fromarray(img).convert("L").save(buf, "JPEG", quality=80)
    data = buf.getvalue()
    assert np.array_equal(JD.decode(data), pillow_rgb(data))


def test_unsupported_files_are_refused():
    buf = io.BytesIO()
    Image.fromarray(photo_like(1, 32, 32)).save(buf, "JPEG", progressive=True)
    with pytest.raises(JD.Unsupported):
        JD.decode(buf.getvalue())
    buf = io.BytesIO()
    Image.fromarray(photo_like(1, 32, 32)).convert("CMYK").save(buf, "JPEG")
    with pytest.raises(JD.Unsupported):
        JD.decode(buf.getvalue())
    with pytest.raises(JD.Unsupported):
        JD.decode(b"not a jpeg")


def _segments_walk(raw):
    """The byte-by-byte statement of jpeg_decode._segments (FF 00 -> FF, FF D0..D7 separate restart segments)."""
    out, cur, i = [], bytearray(), 0
    while i < len(raw):
        b = raw[i]
        if b == 0xFF and i + 1 < len(raw) and raw[i + 1] == 0x00:
            cur.append(0xFF); i += 2
        elif b == 0xFF and i + 1 < len(raw) and 0xD0 <= raw[i + 1] <= 0xD7:
            out.append(bytes(cur)); cur = bytearray(); i += 2
        else:
            cur.append(b); i += 1
    out.append(bytes(cur))
    return out


def test_host_unstuffing_and_scan_end_equal_the_bytewise_walk():
    """The product's host side finds the end of the scan and removes the byte stuffing with re / bytes.replace; both must
    agree with a byte-by-byte walk on streams dense in 0xFF, stuffed zeros and restart markers (host logic, no GPU)."""
    import ctypes as C
    from imagetransformations_amd import jpeg_decode as J, _ffi as F
    rng = np.random.default_rng(11)
    alphabet = np.array([0xFF, 0x00, 0xD0, 0xD3, 0xD7, 0x12, 0xFE, 0xD8, 0xCF], np.uint8)
    for trial in range(300):
        n = int(rng.integers(0, 60))
        raw = bytearray(alphabet[rng.integers(0, len(alphabet), n)].tobytes())
        i = 0
        while i < len(raw):                                   # make it a legal scan body: every FF is followed by 00 or RSTn (or ends the data)
            if raw[i] == 0xFF and i + 1 < len(raw):
                if not (raw[i + 1] == 0x00 or 0xD0 <= raw[i + 1] <= 0xD7):
                    raw[i + 1] = 0x00 if rng.integers(0, 2) else 0xD0 + int(rng.integers(0, 8))
                i += 2
            else:
                i += 1
        raw = bytes(raw)
        assert J._segments(raw) == _segments_walk(raw), raw.hex()
        # the scan inside a file: parse() must stop at the first other marker, or take everything up to the end of the data
        a = np.zeros((8, 8, 3), np.uint8)
        buf = io.BytesIO(); Image.fromarray(a).save(buf, "JPEG")
        f = buf.getvalue()
        sos_end = J.parse(f)["ecs"][0]
        for tail in (b"\xff\xd9", b"\xff\xff\xd9", b"", b"\xff"):
            if raw.endswith(b"\xff"):
                continue                                      # (a trailing FF would pair up with the tail: not a legal scan body)
            g = f[:sos_end] + raw + tail
            s0, s1 = J.parse(g)["ecs"]
            assert (s0, s1) == (sos_end, sos_end + len(raw) + (1 if tail == b"\xff" else 0)), (raw.hex(), tail)
            # the library's host walk (imgxf_jpeg_unstuff_host; no device work): same segments, same end, padded and aligned
            want = _segments_walk(g[s0:s1])
            for keep in (len(want), max(1, len(want) - 1)):
                cap = len(g) + 32 * (keep + 1)
                scan = np.full(cap, 0xAA, np.uint8); off = np.zeros(keep, np.int64); ln = np.zeros(keep, np.int32)
                pos, ns, end = C.c_size_t(16), C.c_int(0), C.c_size_t(0)
                F.call("imgxf_jpeg_unstuff_host", g, len(g), s0, scan.ctypes.data, cap, C.addressof(pos), off.ctypes.data, ln.ctypes.data,
                       keep, C.addressof(ns), C.addressof(end))
                assert ns.value == keep and end.value == s1, (raw.hex(), tail, keep)
                for k in range(keep):
                    assert off[k] % 16 == 0 and scan[off[k]:off[k] + ln[k]].tobytes() == want[k], (raw.hex(), k)
                    assert not scan[off[k] + ln[k]:off[k] + ln[k] + 16].any()
                assert pos.value % 16 == 0 and pos.value <= cap


def _layout(files):
    """imgxf_jpeg_layout_host on a batch (host code of the library; no device work)."""
    import ctypes as C
    from imagetransformations_amd import jpeg_decode as J, _ffi as F
    n = len(files)
    ptrs = (C.c_char_p * n)(*files); sizes = (C.c_size_t * n)(*map(len, files)); status = (C.c_int32 * n)()
    nl, nq, ns = C.c_int(0), C.c_int(0), C.c_int(0)
    sb, ct, pt = C.c_size_t(0), C.c_int64(0), C.c_int64(0)
    F.call("imgxf_jpeg_layout_host", ptrs, sizes, n, None, None, 0, C.addressof(nl), None, 0, C.addressof(nq), None, 0, C.addressof(sb),
           None, None, 0, C.addressof(ns), None, None, status)
    if any(status):
        return list(status), None
    images = (J.DecImage * n)(); luts = (J.DecLut * max(1, nl.value))()
    quants = np.zeros((max(1, nq.value), 64), np.uint16); scan = np.zeros(max(64, sb.value), np.uint8)
    off = np.zeros(max(1, ns.value), np.int64); ln = np.zeros(max(1, ns.value), np.int32)
    F.call("imgxf_jpeg_layout_host", ptrs, sizes, n, images, luts, len(luts), C.addressof(nl), quants.ctypes.data, len(quants), C.addressof(nq),
           scan.ctypes.data, len(scan), C.addressof(sb), off.ctypes.data, ln.ctypes.data, len(off), C.addressof(ns), C.addressof(ct), C.addressof(pt), status)
    return list(status), dict(images=images, luts=luts, n_luts=nl.value, quants=quants, scan=scan, off=off, len=ln, n_segs=ns.value,
                              coef_total=ct.value, plane_total=pt.value)


def test_c_host_layout_equals_the_python_statement():
    """The batch reader's host half (csrc/jpeg_layout.hip) against parse / derive_lut / _segments on the files the reference
    wrote and on seeded files of every sampling, table kind and restart layout; refusals carry the right reason."""
    from imagetransformations_amd import jpeg_decode as J
    files = [open(p, "rb").read() for p in REF[:6]]
    for i, (h, w, kw) in enumerate([(33, 47, dict(subsampling=0)), (64, 80, dict(subsampling=1, optimize=True)), (100, 75, dict(subsampling=2, quality=30)),
                                    (120, 160, dict(restart_marker_rows=2)), (90, 90, dict(restart_marker_blocks=5, quality=95)), (8, 8, {}), (1, 1, {})]):
        buf = io.BytesIO(); Image.fromarray(photo_like(200 + i, h, w)).save(buf, "JPEG", **kw); files.append(buf.getvalue())
    buf = io.BytesIO(); Image.fromarray(photo_like(9, 70, 50)).convert("L").save(buf, "JPEG"); files.append(buf.getvalue())
    status, L = _layout(files)
    assert not any(status)
    seg_next = coef_next = 0
    for i, f in enumerate(files):
        info, im = J.parse(f), L["images"][i]
        comps = info["comps"] if len(info["comps"]) == 3 else [(info["comps"][0][0], 1, 1, info["comps"][0][3])]
        assert (im.width, im.height, im.ncomp) == (info["width"], info["height"], len(comps))
        hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
        assert (im.hmax, im.vmax, im.mcux, im.mcuy) == (hmax, vmax, -(-im.width // (8 * hmax)), -(-im.height // (8 * vmax)))
        total = im.mcux * im.mcuy
        ri = info["dri"] or total
        segs = J._segments(f[info["ecs"][0]:info["ecs"][1]])[: -(-total // ri)]
        assert (im.restart_interval, im.seg_first, im.seg_count) == (ri, seg_next, len(segs))
        for k, sg in enumerate(segs):
            o, n_ = int(L["off"][seg_next + k]), int(L["len"][seg_next + k])
            assert o % 16 == 0 and L["scan"][o:o + n_].tobytes() == sg
        seg_next += len(segs)
        for c, (cid, ch, cv, tq) in enumerate(comps):
            cp = im.comp[c]
            assert (cp.h, cp.v, cp.blocks_x, cp.blocks_y) == (ch, cv, im.mcux * ch, im.mcuy * cv)
            assert (cp.dw, cp.dh) == (-(-im.width * ch // hmax), -(-im.height * cv // vmax))
            assert cp.coef_off == coef_next and cp.plane_off == coef_next
            coef_next += cp.blocks_x * cp.blocks_y * 64
            assert np.array_equal(L["quants"][cp.quant], info["qt"][tq])
            _, td, ta = info["scan"][c]
            for idx, key in ((cp.dc_tab, (0, td)), (cp.ac_tab, (1, ta))):
                assert 0 <= idx < L["n_luts"]
                assert bytes(L["luts"][idx]) == bytes(J.derive_lut(*info["huff"][key]))
    assert L["n_segs"] == seg_next and L["coef_total"] == coef_next == L["plane_total"]
    assert L["n_luts"] <= 4 * len(files)                      # equal tables are shared (the standard tables: 4 for the whole batch)
    # refusals
    good = files[6]
    prog = io.BytesIO(); Image.fromarray(photo_like(1, 64, 64)).save(prog, "JPEG", progressive=True)
    cmyk = io.BytesIO(); Image.fromarray(photo_like(1, 64, 64)).convert("CMYK").save(cmyk, "JPEG")
    status, _ = _layout([good, b"not a jpeg", prog.getvalue(), cmyk.getvalue(), good[:30]])
    assert status == [0, 1, 4, 5, 2]
    rst = files[9]                                            # restart_marker_rows=2: cut after the first segment
    ri = J.parse(rst)
    first_rst = rst.index(b"\xff\xd0", ri["ecs"][0])
    status, _ = _layout([rst[:first_rst] + b"\xff\xd9"])
    assert status == [11]                                     # the scan ends before its last restart segment


def test_c_host_layout_survives_every_truncation_and_byte_flips():
    """The host half reads untrusted files: every prefix of small files and seeded byte flips go through both passes; a file
    is accepted or refused with a reason, the neighbour in the batch is unaffected (tools/fuzz_jpeg_layout.py runs the same
    under AddressSanitizer)."""
    rng = np.random.default_rng(8)
    base = []
    for i, kw in enumerate([dict(), dict(subsampling=0, optimize=True), dict(restart_marker_rows=1)]):
        buf = io.BytesIO(); Image.fromarray(photo_like(300 + i, 24, 40)).save(buf, "JPEG", **kw); base.append(buf.getvalue())
    for f in base:
        cases = [f[:k] for k in range(0, len(f), 2)]
        for _ in range(150):
            g = bytearray(f)
            for _ in range(int(rng.integers(1, 4))):
                g[int(rng.integers(0, len(g)))] = int(rng.integers(0, 256))
            cases.append(bytes(g))
        for g in cases:
            status, _ = _layout([g, base[0]])
            assert status[1] == 0 and 0 <= status[0] <= 11
