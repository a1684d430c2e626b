"""Host side of the device JPEG reader (`libimgxf.so: imgxf_jpeg_decode_*`): the load step of the reference,
`Image.open(path).convert("RGB")` (/root/reference/transformation.py:83; fall_2025/TTA_transforms.py:16-36), for the
files that step meets — baseline / extended-sequential Huffman JPEG, 8 bit, grayscale or YCbCr with 4:4:4, 4:2:2 or
4:2:0 sampling.  The host does what is byte-serial and tiny: it walks the marker segments (jdmarker.c), removes the
byte stuffing of the scan, splits it at RSTn markers, derives the decoding tables of the Huffman specifications
(jdhuff.c jpeg_make_d_derived_tbl) and lays the batch out; entropy decoding, dequantisation + IDCT, upsampling and
colour conversion run on the device (csrc/jpeg_decode.hip) and produce the pixels Pillow / libjpeg-turbo produces, bit
for bit.  Nothing here computes a pixel; a file outside that class raises `UnsupportedJpeg` — there is no CPU fallback
in this module (io_pipeline decides what to do with such a file)."""
from __future__ import annotations

import ctypes as C
import re
from typing import List, Sequence

import numpy as np
import torch

from . import _ffi as F

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14,
                   21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53,
                   60, 61, 54, 47, 55, 62, 63])


class UnsupportedJpeg(ValueError):
    """Not a file the device reader covers (progressive, arithmetic, 12-bit, CMYK, non-interleaved scans, ...)."""


class DecComp(C.Structure):
    """struct imgxf_jpeg_dec_comp (include/imgxf.h)"""
    _fields_ = [("h", C.c_int32), ("v", C.c_int32), ("dc_tab", C.c_int32), ("ac_tab", C.c_int32), ("quant", C.c_int32),
                ("blocks_x", C.c_int32), ("blocks_y", C.c_int32), ("dw", C.c_int32), ("dh", C.c_int32),
                ("coef_off", C.c_int64), ("plane_off", C.c_int64)]


class DecImage(C.Structure):
    """struct imgxf_jpeg_dec_image"""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("ncomp", C.c_int32), ("hmax", C.c_int32), ("vmax", C.c_int32),
                ("mcux", C.c_int32), ("mcuy", C.c_int32), ("restart_interval", C.c_int32), ("seg_first", C.c_int32),
                ("seg_count", C.c_int32), ("pad_", C.c_int32), ("out_off", C.c_int64), ("out_pitch", C.c_int64),
                ("comp", DecComp * 3)]


class DecLut(C.Structure):
    """struct imgxf_jpeg_dec_lut"""
    _fields_ = [("look", C.c_uint16 * 256), ("maxcode", C.c_int32 * 18), ("valoff", C.c_int32 * 17), ("huffval", C.c_uint8 * 256)]


def parse(data: bytes, find_end: bool = True) -> dict:
    """Marker segments up to the scan (jdmarker.c): frame, tables, restart interval, the entropy-coded byte range
    (`find_end=False`: ecs = (start, None); the batched reader gets the end from imgxf_jpeg_unstuff_host's walk)."""
    if len(data) < 4 or data[0] != 0xFF or data[1] != 0xD8:
        raise UnsupportedJpeg("not a JPEG (no SOI)")
    pos, qt, huff, frame, dri = 2, {}, {}, None, 0
    n = len(data)
    while True:
        if pos + 4 > n or data[pos] != 0xFF:
            raise UnsupportedJpeg("damaged marker structure")
        while data[pos + 1] == 0xFF and pos + 2 < n:
            pos += 1
        marker = data[pos + 1]
        seglen = (data[pos + 2] << 8) | data[pos + 3]
        seg = data[pos + 4:pos + 2 + seglen]
        if marker == 0xDB:
            i = 0
            while i < len(seg):
                pq, tq = seg[i] >> 4, seg[i] & 15
                i += 1
                t = np.zeros(64, np.uint16)
                if pq:
                    t[ZIGZAG] = np.frombuffer(seg[i:i + 128], ">u2")
                    i += 128
                else:
                    t[ZIGZAG] = np.frombuffer(seg[i:i + 64], np.uint8)
                    i += 64
                qt[tq] = t
        elif marker in (0xC0, 0xC1):
            if seg[0] != 8:
                raise UnsupportedJpeg(f"{seg[0]}-bit samples")
            h, w, nc = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4], seg[5]
            frame = (w, h, [(seg[6 + 3 * k], seg[7 + 3 * k] >> 4, seg[7 + 3 * k] & 15, seg[8 + 3 * k]) for k in range(nc)])
        elif 0xC2 <= marker <= 0xCF and marker not in (0xC4, 0xC8, 0xCC):
            raise UnsupportedJpeg(f"SOF{marker - 0xC0}: progressive, lossless or arithmetic coding")
        elif marker == 0xC4:
            i = 0
            while i < len(seg):
                tc, th = seg[i] >> 4, seg[i] & 15
                bits = tuple(seg[i + 1:i + 17])
                cnt = sum(bits)
                huff[(tc, th)] = (bits, bytes(seg[i + 17:i + 17 + cnt]))
                i += 17 + cnt
        elif marker == 0xDD:
            dri = (seg[0] << 8) | seg[1]
        elif marker == 0xDA:
            if frame is None:
                raise UnsupportedJpeg("SOS before SOF")
            w, h, comps = frame
            ns = seg[0]
            if len(comps) not in (1, 3) or ns != len(comps):
                raise UnsupportedJpeg(f"{len(comps)} components in {ns}-component scans")
            ids = [c[0] for c in comps]
            scan = []
            for k in range(ns):
                cid, tt = seg[1 + 2 * k], seg[2 + 2 * k]
                if cid not in ids:
                    raise UnsupportedJpeg("scan names an unknown component")
                scan.append((ids.index(cid), tt >> 4, tt & 15))
            if [s[0] for s in scan] != list(range(ns)):
                raise UnsupportedJpeg("scan components out of frame order")
            start = pos + 2 + seglen
            end = None
            if find_end:
                m = _ECS_END.search(data, start)            # (a lone 0xFF as the file's last byte belongs to the scan)
                end = m.start() if m else n
            return dict(width=w, height=h, comps=comps, qt=qt, huff=huff, scan=scan, dri=dri, ecs=(start, end))
        pos += 2 + seglen


_LUT_CACHE: dict = {}


def derive_lut(bits: Sequence[int], vals: bytes) -> DecLut:
    """jdhuff.c jpeg_make_d_derived_tbl: 8-bit lookahead + maxcode / valoff for the longer codes."""
    key = (tuple(bits), bytes(vals))
    lut = _LUT_CACHE.get(key)
    if lut is not None:
        return lut
    lut = DecLut()
    for i in range(18):
        lut.maxcode[i] = -1
    lut.maxcode[17] = 0xFFFFF
    code, k = 0, 0
    for length in range(1, 17):
        cnt = bits[length - 1]
        if cnt:
            lut.valoff[length] = k - code
            for j in range(cnt):
                if length <= 8:
                    first = (code + j) << (8 - length)
                    entry = (length << 8) | vals[k + j]
                    for e in range(first, first + (1 << (8 - length))):
                        lut.look[e] = entry
            k += cnt
            code += cnt
            lut.maxcode[length] = code - 1
        code <<= 1
    for i, v in enumerate(vals[:256]):
        lut.huffval[i] = v
    if len(_LUT_CACHE) < 4096:
        _LUT_CACHE[key] = lut
    return lut


_RST = re.compile(rb"\xff[\xd0-\xd7]")
_ECS_END = re.compile(rb"\xff[^\x00\xd0-\xd7]")           # the first marker that is neither a stuffed zero nor RSTn ends the scan


def _segments(raw: bytes):
    """The entropy-coded bytes of a scan -> restart segments with the byte stuffing removed (FF 00 -> FF; FF D0..D7
    separate segments).  `raw` is parse()'s `ecs` range: it holds no other marker.  bytes.replace / re.split run at C
    speed; a byte-by-byte Python walk over every 0xFF was most of the reader's host time."""
    if b"\xff" not in raw:
        return [raw]
    return [p.replace(b"\xff\x00", b"\xff") for p in _RST.split(raw)]


_REFUSALS = {1: "not a JPEG (no SOI)", 2: "damaged marker structure", 3: "samples are not 8 bits",
             4: "progressive, lossless or arithmetic coding", 5: "neither 1 nor 3 components, or a non-interleaved scan",
             6: "the scan names an unknown component or not in frame order", 7: "sampling factors outside 1..2",
             8: "chroma sampling other than 4:4:4, 4:2:2 (h2v1) or 4:2:0", 9: "missing quantisation table", 10: "missing Huffman table"}


def _raise_for_status(status, n: int) -> None:
    """imgxf_jpeg_layout_host's per-file verdicts -> the exceptions of the Python statement (first refused file)."""
    for i in range(n):
        code = status[i]
        if code == 11:
            raise F.ImgxfError(F.ERR_ARG, f"file {i}: the scan ends before its last restart segment", "jpeg_decode.decode")
        if code:
            raise UnsupportedJpeg(f"file {i}: {_REFUSALS.get(code, code)}")


LAST_PROFILE: dict = {}        # filled by decode(..., profile=True): seconds per stage of the last call (it synchronises)


def decode(files: Sequence[bytes], device=None, profile: bool = False) -> List[torch.Tensor]:
    """One [H, W, 3] uint8 RGB device tensor per file: the pixels of `Image.open(BytesIO(f)).convert("RGB")`.
    Files of equal size share one [N, H, W, 3] allocation (the views are its frames), which is what the batched drivers
    group by.  Raises UnsupportedJpeg for a file outside the reader's class, ImgxfError for a damaged stream."""
    device = torch.device("cuda") if device is None else torch.device(device)
    if device.type != "cuda":
        raise F.ImgxfError(F.ERR_NO_DEVICE, "the JPEG reader runs on the GPU (no CPU fallback)", "jpeg_decode.decode")
    import time
    n = len(files)
    if n == 0:
        return []
    t_start = time.perf_counter()
    files = [bytes(f) for f in files]
    # host half in C (csrc/jpeg_layout.hip: the statement of parse / derive_lut / _segments above for a whole batch)
    ptrs = (C.c_char_p * n)(*files)
    sizes = (C.c_size_t * n)(*map(len, files))
    status_h = (C.c_int32 * n)()
    n_luts, n_quants, n_segs = C.c_int(0), C.c_int(0), C.c_int(0)
    scan_bytes, coef_total, plane_total = C.c_size_t(0), C.c_int64(0), C.c_int64(0)
    F.call("imgxf_jpeg_layout_host", ptrs, sizes, n, None, None, 0, C.addressof(n_luts), None, 0, C.addressof(n_quants), None, 0,
           C.addressof(scan_bytes), None, None, 0, C.addressof(n_segs), None, None, status_h)
    _raise_for_status(status_h, n)
    images = (DecImage * n)()
    lut_cap, quant_cap, seg_cap, scan_cap = max(1, n_luts.value), max(1, n_quants.value), max(1, n_segs.value), max(64, scan_bytes.value)
    lut_arr = (DecLut * lut_cap)()
    quants_h = torch.empty((quant_cap, 64), dtype=torch.int16)
    scan_host = torch.empty((scan_cap,), dtype=torch.uint8)
    seg_off_h = torch.zeros((seg_cap,), dtype=torch.int64)
    seg_len_h = torch.zeros((seg_cap,), dtype=torch.int32)
    F.call("imgxf_jpeg_layout_host", ptrs, sizes, n, images, lut_arr, lut_cap, C.addressof(n_luts), quants_h.data_ptr(), quant_cap,
           C.addressof(n_quants), scan_host.data_ptr(), scan_cap, C.addressof(scan_bytes), seg_off_h.data_ptr(), seg_len_h.data_ptr(), seg_cap,
           C.addressof(n_segs), C.addressof(coef_total), C.addressof(plane_total), status_h)
    _raise_for_status(status_h, n)
    coef_pos, plane_pos = coef_total.value, plane_total.value
    by_size: dict = {}
    for i in range(n):
        by_size.setdefault((images[i].height, images[i].width), []).append(i)

    # one [N, H, W, 3] tensor per size; frames are handed back in file order
    results: List[torch.Tensor] = [None] * n
    out_pos = 0
    spans = []
    for (h, w), members in by_size.items():
        for j, i in enumerate(members):
            images[i].out_off = out_pos + j * h * w * 3
            images[i].out_pitch = w * 3
        spans.append((out_pos, len(members), h, w, members))
        out_pos += (len(members) * h * w * 3 + 15) & ~15
    t_host = time.perf_counter()

    def mark(name, t0):
        if profile:
            torch.cuda.synchronize(device)
            LAST_PROFILE[name] = time.perf_counter() - t0
            return time.perf_counter()
        return t0

    with torch.cuda.device(device):
        if profile:
            LAST_PROFILE.clear()
            LAST_PROFILE["host parse + tables"] = t_host - t_start
        stream = torch.cuda.current_stream(device).cuda_stream
        out = torch.empty((out_pos,), dtype=torch.uint8, device=device)
        scan_d = scan_host[:max(16, scan_bytes.value)].to(device, non_blocking=False)
        seg_off_d = seg_off_h.to(device)
        seg_len_d = seg_len_h.to(device)
        images_d = torch.frombuffer(bytearray(bytes(images)), dtype=torch.uint8).to(device)
        luts_d = torch.frombuffer(bytearray(bytes(lut_arr)), dtype=torch.uint8).to(device)
        quants_d = quants_h.to(device)
        coefs = torch.zeros((coef_pos,), dtype=torch.int16, device=device)
        planes = torch.empty((plane_pos,), dtype=torch.uint8, device=device)
        status = torch.zeros((n,), dtype=torch.int32, device=device)
        t0 = mark("uploads + zero fill", t_host)
        F.call("imgxf_jpeg_decode_huffman", scan_d.data_ptr(), seg_off_d.data_ptr(), seg_len_d.data_ptr(), images_d.data_ptr(), n,
               luts_d.data_ptr(), coefs.data_ptr(), status.data_ptr(), stream)
        t0 = mark("huffman kernel", t0)
        F.call("imgxf_jpeg_decode_idct", coefs.data_ptr(), images_d.data_ptr(), C.addressof(images), n, quants_d.data_ptr(),
               planes.data_ptr(), stream)
        t0 = mark("idct kernel", t0)
        F.call("imgxf_jpeg_decode_color", planes.data_ptr(), images_d.data_ptr(), C.addressof(images), n, out.data_ptr(), stream)
        t0 = mark("upsample + colour kernel", t0)
        bad = torch.nonzero(status).flatten().tolist()
    if bad:
        raise F.ImgxfError(F.ERR_ARG, f"damaged entropy-coded data in file(s) {bad}", "jpeg_decode.decode")
    for off, cnt, h, w, members in spans:
        frames = out[off:off + cnt * h * w * 3].view(cnt, h, w, 3).unbind(0)     # (one call: indexing frame by frame costs 3 us each)
        for j, i in enumerate(members):
            results[i] = frames[j]
    return results


def decode_batches(files: Sequence[bytes], device=None):
    """`decode`, grouped: {(height, width): ([N, H, W, 3] tensor, [file indices])} — the layout the batched drivers use."""
    frames = decode(files, device)
    groups: dict = {}
    for i, t in enumerate(frames):
        groups.setdefault((t.shape[0], t.shape[1]), []).append(i)
    return {k: (torch.stack([frames[i] for i in v]) if len(v) > 1 else frames[v[0]][None], v) for k, v in groups.items()}
