import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def synth(seed, h, w, c=3):
    """Synthetic uint8 image, SURVEY §8c: default_rng(seed).integers(0,256,...)."""
    shape = (h, w) if c == 1 else (h, w, c)
    return np.random.default_rng(seed).integers(0, 256, shape, dtype=np.uint8)


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    return torch.device("cuda:0")


class _KnobAwareMonkeypatch:
    """pytest's monkeypatch, plus: libimgxf caches its IMGXF_* knobs at first use, so a change of
    one inside a test is followed by imgxf_reload_knobs() (and once more at teardown)."""

    def __init__(self, mp):
        self._mp = mp
        self.touched = False

    @staticmethod
    def _is_knob(name):
        return name.startswith("IMGXF_") and not name.startswith(("IMGXF_LIBRARY", "IMGXF_BENCH", "IMGXF_SOAK"))

    def _reload(self):
        mod = sys.modules.get("imagetransformations_amd._ffi")
        if mod is not None:
            mod.reload_knobs()

    def setenv(self, name, value, prepend=None):
        self._mp.setenv(name, value, prepend)
        if self._is_knob(name):
            self.touched = True
            self._reload()

    def delenv(self, name, raising=True):
        self._mp.delenv(name, raising)
        if self._is_knob(name):
            self.touched = True
            self._reload()

    def __getattr__(self, item):
        return getattr(self._mp, item)


@pytest.fixture
def monkeypatch(monkeypatch):
    wrapped = _KnobAwareMonkeypatch(monkeypatch)
    yield wrapped
    monkeypatch.undo()
    if wrapped.touched:
        wrapped._reload()
