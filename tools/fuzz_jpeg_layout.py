"""Truncations and byte flips of small JPEG files through the reader's HOST half (imgxf_jpeg_layout_host, csrc/jpeg_layout.hip),
meant to run against an AddressSanitizer build of that one file on the CPU (development aid):
    g++ -std=c++17 -O1 -g -fsanitize=address,undefined -shared -fPIC -D__HIP_PLATFORM_AMD__ -Iinclude -Iimagetransformations_amd/csrc \
        -I/opt/rocm/include -x c++ imagetransformations_amd/csrc/jpeg_layout.hip -o _exp/asan/libjpeg_layout_asan.so
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tools/fuzz_jpeg_layout.py _exp/asan/libjpeg_layout_asan.so"""
import ctypes as C, io, sys
import numpy as np
from PIL import Image

lib = C.CDLL(sys.argv[1])
fn = lib.imgxf_jpeg_layout_host
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
               C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
IMG, LUT = 232, 908                                      # sizeof(imgxf_jpeg_dec_image), sizeof(imgxf_jpeg_dec_lut)


def layout(files):
    n = len(files)
    # exact-size heap copies, so that a read past the end of a file is a heap overflow the sanitizer sees
    bufs = [(C.c_uint8 * max(1, len(f))).from_buffer_copy(f if f else b"\0") for f in files]
    ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs]); sizes = (C.c_size_t * n)(*map(len, files)); status = (C.c_int32 * n)()
    nl, nq, ns = C.c_int(0), C.c_int(0), C.c_int(0)
    sb, ct, pt = C.c_size_t(0), C.c_int64(0), C.c_int64(0)
    rc = fn(ptrs, sizes, n, None, None, 0, C.addressof(nl), None, 0, C.addressof(nq), None, 0, C.addressof(sb), None, None, 0, C.addressof(ns), None, None, status)
    assert rc == 0
    images = (C.c_uint8 * (IMG * n))(); luts = (C.c_uint8 * (LUT * max(1, nl.value)))()
    quants = (C.c_uint16 * (64 * max(1, nq.value)))(); scan = (C.c_uint8 * max(64, sb.value))()
    off = (C.c_int64 * max(1, ns.value))(); ln = (C.c_int32 * max(1, ns.value))()
    rc = fn(ptrs, sizes, n, images, luts, max(1, nl.value), C.addressof(nl), quants, max(1, nq.value), C.addressof(nq), scan, max(64, sb.value),
            C.addressof(sb), off, ln, max(1, ns.value), C.addressof(ns), C.addressof(ct), C.addressof(pt), status)
    assert rc == 0, rc
    return list(status)


rng = np.random.default_rng(5)
yy, xx = np.mgrid[0:40, 0:48]
seeds = []
for i, kw in enumerate([dict(), dict(subsampling=0), dict(optimize=True, subsampling=1), dict(restart_marker_rows=1), dict(quality=98)]):
    img = np.clip((128 + 60 * np.sin(xx / 7.0) + rng.normal(0, 9, (40, 48)))[..., None] + np.zeros(3), 0, 255).astype(np.uint8)
    b = io.BytesIO(); Image.fromarray(img).save(b, "JPEG", **kw); seeds.append(b.getvalue())
b = io.BytesIO(); Image.fromarray(img).convert("L").save(b, "JPEG"); seeds.append(b.getvalue())
cases = accepted = 0
for f in seeds:
    todo = [f[:k] for k in range(0, len(f) + 1)]
    for _ in range(1500):
        g = bytearray(f)
        for _ in range(int(rng.integers(1, 5))):
            g[int(rng.integers(0, len(g)))] = int(rng.integers(0, 256))
        todo.append(bytes(g))
    for g in todo:
        st = layout([g, seeds[0]])
        cases += 1; accepted += st[0] == 0
        assert st[1] == 0
print("cases", cases, "accepted", accepted, "- no sanitizer report")
