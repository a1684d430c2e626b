"""CPU-side checks of the drop-in boundary: libimgxf.so loads, exports every symbol that
include/imgxf.h declares, and the ctypes table in _ffi.py matches the header.  No compute."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
HEADER = (ROOT / "include" / "imgxf.h").read_text()


def declared_functions():
    # strip comments, then find `imgxf_name(` at declaration level
    text = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    return sorted(set(re.findall(r"\b(imgxf_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_hot_path_entry_points():
    names = declared_functions()
    for must in ("imgxf_gaussian_u8", "imgxf_conv2d_u8", "imgxf_sobel_u8", "imgxf_affine_u8",
                 "imgxf_resize_lanczos_u8", "imgxf_rgb2l_u8", "imgxf_scale_abs_u8", "imgxf_blend_u8",
                 "imgxf_add_noise_u8", "imgxf_permute_u8"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from imagetransformations_amd import _ffi
    lib = ctypes.CDLL(str(_ffi.LIB_PATH))
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"declared in imgxf.h but not exported: {missing}"


def test_ctypes_table_covers_the_header():
    from imagetransformations_amd import _ffi
    bound = set(_ffi.SIGNATURES) | {"imgxf_strerror"}
    assert set(declared_functions()) == bound


def test_version_and_error_strings():
    from imagetransformations_amd import _ffi
    assert _ffi.lib.imgxf_version() == 100
    assert _ffi.strerror(0) == "ok"
    for code in (-1, -2, -3, -4, -5, -6):
        assert _ffi.strerror(code).startswith("imgxf:")
    with pytest.raises(ValueError):
        _ffi.check(_ffi.ERR_SHAPE, "x")
    with pytest.raises(_ffi.ImgxfError):
        _ffi.check(_ffi.ERR_UNSUPPORTED, "x")


def test_argument_validation_needs_no_gpu():
    """NULL / malformed views are rejected on the host before any launch."""
    from imagetransformations_amd import _ffi
    v = _ffi.View(None, 1, 4, 4, 3, 12, 48)
    assert _ffi.lib.imgxf_gaussian_u8(ctypes.byref(v), ctypes.byref(v), 5, 1.0, None, None) == _ffi.ERR_NULL
    bad = _ffi.View(1, 1, 4, 4, 3, 5, 48)            # row_stride < w*c
    assert _ffi.lib.imgxf_rgb2l_u8(ctypes.byref(bad), ctypes.byref(bad), None) == _ffi.ERR_SHAPE
    ok = _ffi.View(16, 1, 4, 4, 3, 12, 48)
    assert _ffi.lib.imgxf_gaussian_u8(ctypes.byref(ok), ctypes.byref(ok), 4, 1.0, None, None) == _ffi.ERR_ARG


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    import importlib
    import sys
    monkeypatch.setenv("IMGXF_LIBRARY", str(tmp_path / "nope.so"))
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k.startswith("imagetransformations_amd")}
    try:
        with pytest.raises(ImportError, match="no CPU fallback"):
            importlib.import_module("imagetransformations_amd")
    finally:
        for k in list(sys.modules):
            if k.startswith("imagetransformations_amd"):
                sys.modules.pop(k)
        sys.modules.update(saved)
