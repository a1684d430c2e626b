"""Build libimgxf.so (hand-written HIP, gfx950 only) in-tree with hipcc.

`python -m imagetransformations_amd.build` or `build_library()`; hipcc cross-compiles
without a GPU.  Objects are cached under csrc/_obj and rebuilt when a source or header is
newer.  The resulting `imagetransformations_amd/libimgxf.so` is git-ignored but travels to
the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
INCLUDE = PKG.parent / "include"
OBJ = CSRC / "_obj"
LIB = PKG / "libimgxf.so"
ARCH = "gfx950"
FLAGS = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", "-fvisibility=hidden",
         # IEEE sequencing everywhere: Pillow/NumPy parity needs un-fused mul+add; kernels that
         # want an FMA say fmaf() explicitly.
         "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", f"-I{INCLUDE}", f"-I{CSRC}"]


# per-file additions.  affine.hip: hipcc's SLP vectoriser turns the bilinear kernels' 96 scalar v_fmac_f32 per 8 pixels into
# 48 v_pk_fma_f32 across neighbouring pixels — fewer instructions, but a packed FMA issues in 4.5 cycles against 2.8 for a
# scalar one, and the batch kernels measured 2 - 6 % SLOWER with it (profiles/r03_experiments/ab_affine_slp.txt)
FILE_FLAGS = {"affine.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libimgxf.so cannot be built")
    return exe


def _newest_header() -> float:
    deps = list(CSRC.glob("*.h")) + list(CSRC.glob("*.inc")) + list(INCLUDE.glob("*.h"))
    return max(p.stat().st_mtime for p in deps)


def build_library(force: bool = False, verbose: bool = True, jobs: int | None = None) -> Path:
    hipcc = _hipcc()
    OBJ.mkdir(exist_ok=True)
    sources = sorted(CSRC.glob("*.hip"))
    hdr_time = _newest_header()
    stamp = OBJ / "flags.txt"
    flag_text = " ".join(FLAGS) + " | " + repr(sorted(FILE_FLAGS.items()))
    if not stamp.exists() or stamp.read_text() != flag_text:
        force = True
        stamp.write_text(flag_text)
    todo = []
    objs = []
    for src in sources:
        obj = OBJ / (src.stem + ".o")
        objs.append(obj)
        if force or not obj.exists() or obj.stat().st_mtime < max(src.stat().st_mtime, hdr_time):
            todo.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc, *FLAGS, *FILE_FLAGS.get(src.name, []), "-c", str(src), "-o", str(obj)]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{res.stdout}\n{res.stderr}")
        if verbose and res.stderr.strip():
            print(res.stderr, file=sys.stderr)
        return src.name

    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
            for name in ex.map(compile_one, todo):
                if verbose:
                    print(f"[imgxf build] compiled {name}")
    if todo or not LIB.exists():
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
        if verbose:
            print(f"[imgxf build] linked {LIB}")
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
