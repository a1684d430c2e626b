"""imagetransformations_amd — MI355X-native (gfx950, hand-written HIP) implementation of the
per-pixel transform hot path of aaryaamoharir/ImageTransformations.

    from imagetransformations_amd import transformation as T   # drop-in apply_* functions
    from imagetransformations_amd import ops                    # batched device-tensor API

Importing the package loads libimgxf.so through ctypes and raises ImportError if it has
not been built (`python -m imagetransformations_amd.build`).  There is no CPU fallback.
"""
import sys as _sys

# `python -m imagetransformations_amd.build` must be able to (re)build a missing or stale
# library, so that one entry point skips the eager load; every other import fails loudly.
_building = "imagetransformations_amd.build" in getattr(_sys, "orig_argv", [])
if not _building:
    from . import _ffi  # noqa: F401  (raises ImportError / AttributeError when the HIP library is missing or stale)

__version__ = "0.1.0"
__all__ = ["ops", "transformation", "transformations_code", "pool", "augmix", "tensor_maps", "io_pipeline", "sharding", "_ffi"]


def __getattr__(name):
    if name in ("ops", "transformation", "transformations_code", "sharding", "pool", "augmix", "tensor_maps", "io_pipeline"):
        import importlib
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
