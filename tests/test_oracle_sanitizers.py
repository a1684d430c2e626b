"""CPU: the C restatement of the oracle (oracle/c/imgxf_oracle.c) built with AddressSanitizer +
UndefinedBehaviorSanitizer and driven over the same cases as the NumPy cross-check (SURVEY §5:
GPU sanitizers are not available on this pool, so the CPU build is the one that runs under them).
The instrumented library runs in a child process (ASan wants to be first in the link order)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import ctypes as C, sys, numpy as np
sys.path.insert(0, sys.argv[2]); sys.path.insert(0, sys.argv[2] + "/tests")
from oracle import c_oracle as CO, imgxf_oracle as O
CO._lib = C.CDLL(sys.argv[1])                       # the sanitizer build instead of oracle/_build
rng = np.random.default_rng(3)
for (h, w) in ((1, 1), (3, 5), (37, 61), (48, 64), (64, 200)):
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    for k, s in ((3, 0.5), (5, 5 / 6), (13, 2.0), (31, 5.0)):
        got, want = CO.gaussian_blur(a, k, s), O.gaussian_blur(a, k, s)
        assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
    for m in (O.rotate_zoom_matrix(w, h, 30.0, 1.5), O.rotate_zoom_matrix(w, h, -100.0, 0.7)):
        assert np.array_equal(CO.affine(a, (w, h), m, 1, (0, 0, 0)), O.affine_bilinear(a, (w, h), m, fill=(0, 0, 0)))
        assert np.array_equal(CO.affine(a, (w + 7, h + 3), m, 1, (1, 2, 3)), O.affine_bilinear(a, (w + 7, h + 3), m, fill=(1, 2, 3)))
CO.set_threads(4)
a = rng.integers(0, 256, (270, 480, 3), dtype=np.uint8)
assert np.abs(CO.gaussian_blur(a, 5, 5 / 6).astype(int) - O.gaussian_blur(a, 5, 5 / 6).astype(int)).max() <= 1
print("sanitized oracle ok")
'''


def test_c_oracle_under_asan_and_ubsan(tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    so = str(tmp_path / "libimgxf_oracle_san.so")
    src = os.path.join(ROOT, "oracle", "c", "imgxf_oracle.c")
    cmd = [gcc, "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           "-march=x86-64-v3", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", "-std=c11", src, "-o", so, "-lm"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0 and "asan" in (res.stderr or "").lower():
        pytest.skip("this gcc has no sanitizer runtime: " + res.stderr[-200:])
    assert res.returncode == 0, res.stderr
    libasan = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan.so not found")
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", OMP_NUM_THREADS="4")
    run = subprocess.run([sys.executable, "-c", SCRIPT, so, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "sanitized oracle ok" in run.stdout, (run.stdout[-500:], run.stderr[-3000:])
