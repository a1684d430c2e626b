"""GPU: libimgxf.so called from a plain C program (tests/c_abi/c_abi_check.c) — no Python, no
torch: hipMalloc'd buffers, a caller-created stream — and checked inside that program against
the C oracle.  The binary is compiled here with gcc against the HIP runtime headers."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_program_through_the_c_abi(device, tmp_path):
    from oracle import c_oracle
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.isdir("/opt/rocm/include/hip"):
        pytest.skip("gcc or the HIP headers are not available")
    oracle_so = str(c_oracle.build())
    lib_dir = os.path.join(ROOT, "imagetransformations_amd")
    assert os.path.exists(os.path.join(lib_dir, "libimgxf.so")), "libimgxf.so missing: run __graft_entry__.build()"
    exe = str(tmp_path / "c_abi_check")
    cmd = [gcc, "-O1", "-std=c11", "-D__HIP_PLATFORM_AMD__", "-D_GNU_SOURCE", f"-I{ROOT}/include", "-I/opt/rocm/include",
           os.path.join(ROOT, "tests", "c_abi", "c_abi_check.c"), "-o", exe,
           f"-L{lib_dir}", "-limgxf", oracle_so, "-L/opt/rocm/lib", "-lamdhip64", "-lm",
           f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{os.path.dirname(oracle_so)}", "-Wl,-rpath,/opt/rocm/lib"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "c_abi ok" in run.stdout, (run.returncode, run.stdout, run.stderr)
