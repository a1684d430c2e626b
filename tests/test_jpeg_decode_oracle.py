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
