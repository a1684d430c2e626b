"""Build + ctypes loader for oracle/c/imgxf_oracle.c (TEST INFRASTRUCTURE ONLY).

Used by tests/ (cross-check of the NumPy oracle) and by bench.py's `cpu_baseline` leg.
The product package never imports this.  The shared object lands in oracle/_build/
(git-ignored, but it travels to the GPU box with the repo snapshot)."""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
SRC = HERE / "c" / "imgxf_oracle.c"
OUT = HERE / "_build" / "libimgxf_oracle_c.so"
# x86-64-v3 (AVX2) is safe on any host this runs on; contraction off keeps IEEE sequencing
CFLAGS = ["-O3", "-march=x86-64-v3", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", "-std=c11"]


def build(force: bool = False) -> Path:
    gcc = shutil.which("gcc")
    if gcc is None:
        raise RuntimeError("gcc not found: cannot build the C oracle")
    if force or not OUT.exists() or OUT.stat().st_mtime < SRC.stat().st_mtime:
        OUT.parent.mkdir(exist_ok=True)
        subprocess.run([gcc, *CFLAGS, str(SRC), "-o", str(OUT), "-lm"], check=True)
    return OUT


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(str(build()))
    return _lib


def _p(a, t=C.c_uint8):
    return a.ctypes.data_as(C.POINTER(t))


def set_threads(n: int) -> int:
    return lib().oracle_set_threads(int(n))


def gaussian_blur(img: np.ndarray, ksize: int, sigma: float, out=None, tmp=None) -> np.ndarray:
    """`out` / `tmp` (float64, img.size): optional pre-allocated buffers (the timed CPU baseline reuses them
    so that page faults of fresh 200 MB temporaries are not what it measures)."""
    a = np.ascontiguousarray(img)
    h, w = a.shape[:2]
    c = 1 if a.ndim == 2 else a.shape[2]
    out = np.empty_like(a) if out is None else out
    tmp = np.empty(a.size, np.float64) if tmp is None else tmp
    rc = lib().oracle_gaussian_blur_u8(_p(a), _p(out), _p(tmp, C.c_double), h, w, c, int(ksize), C.c_double(sigma))
    if rc:
        raise ValueError(f"oracle_gaussian_blur_u8 -> {rc}")
    return out


def affine(img: np.ndarray, out_size, m, filter: int, fill=None, out=None) -> np.ndarray:
    a = np.ascontiguousarray(img)
    h, w = a.shape[:2]
    c = 1 if a.ndim == 2 else a.shape[2]
    ow, oh = out_size
    out = np.empty((oh, ow) if a.ndim == 2 else (oh, ow, c), np.uint8) if out is None else out
    mm = (C.c_double * 6)(*[float(v) for v in m])
    ff = (C.c_uint8 * 4)(*(list(fill)[:c] + [0] * (4 - min(c, len(fill))))) if fill is not None else None
    rc = lib().oracle_affine_u8(_p(a), h, w, c, _p(out), oh, ow, mm, int(filter), ff)
    if rc:
        raise ValueError(f"oracle_affine_u8 -> {rc}")
    return out


def rgb2l(img: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(img)
    h, w, c = a.shape
    out = np.empty((h, w), np.uint8)
    lib().oracle_rgb2l_u8(_p(a), _p(out), h, w, c)
    return out


def sobel(gray: np.ndarray, variant: int) -> np.ndarray:
    g = np.ascontiguousarray(gray)
    out = np.empty_like(g)
    lib().oracle_sobel_u8(_p(g), _p(out), g.shape[0], g.shape[1], int(variant))
    return out
