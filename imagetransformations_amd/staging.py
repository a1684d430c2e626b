"""Pinned host staging for the PIL / NumPy edges of the hot path.

The reference reads an image, transforms it and saves it, one call at a time
(/root/reference/transformation.py:83 `Image.open().convert("RGB")`, :161-162 `save`).  Here the
pixels cross PCIe: pageable `tensor.to(device)` / `tensor.cpu()` make the driver stage through its
own bounce buffer and block the host for the whole copy.  This module keeps the copies
asynchronous on the current stream through PINNED memory from torch's caching host allocator
(a freed pinned block is reused, not re-registered, so steady-state calls allocate nothing):

    upload(arrays)      np arrays -> one pinned [N,...] block (one host memcpy) -> async H2D
    download(t)         async D2H into a pinned block; .numpy() waits for THAT copy only

so a batched driver can queue group k+1's upload and group k's launches while group k-1's
results are still on their way back, and synchronise once per result when it builds the images.
"""
from __future__ import annotations

import os
import weakref
from typing import Sequence

import numpy as np
import torch


PENDING_BUDGET = 1 << 30   # bytes of asynchronous copies back a batched driver keeps in flight (pinned memory is
                           # never returned by torch's caching host allocator: the window bounds the footprint)
SMALL = 256 * 1024      # below this the driver's own small-copy path beats a pinned block + event (32x32 CIFAR images)


def upload(arrays, device: torch.device) -> torch.Tensor:
    """np.ndarray or a sequence of equally shaped arrays -> device tensor ([N, ...] for a sequence)."""
    if isinstance(arrays, np.ndarray) and arrays.nbytes < SMALL:
        return torch.from_numpy(arrays).to(device)
    if isinstance(arrays, np.ndarray):
        shape, dtype, seq = arrays.shape, arrays.dtype, None
    else:
        seq = list(arrays)
        shape, dtype = (len(seq),) + tuple(seq[0].shape), seq[0].dtype
    pinned = torch.empty(shape, dtype=torch.from_numpy(np.empty(0, dtype)).dtype, pin_memory=True)
    host = pinned.numpy()
    if seq is None:
        np.copyto(host, arrays)
    else:
        for i, a in enumerate(seq):
            np.copyto(host[i], a)
    # the caching host allocator ties the block to the stream of this copy: it is not handed out
    # again before the copy has run, although `pinned` is dropped right here
    return pinned.to(device, non_blocking=True)


class Download:
    """An asynchronous device -> pinned-host copy; `numpy()` waits for it (and only for it)."""

    def __init__(self, t: torch.Tensor):
        self._done = None
        if t.numel() * t.element_size() < SMALL:
            self._host = t.cpu()
            return
        self._host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        with torch.cuda.device(t.device):
            self._host.copy_(t, non_blocking=True)
            self._done = torch.cuda.Event()
            self._done.record(torch.cuda.current_stream(t.device))

    def numpy(self) -> np.ndarray:
        if self._done is not None:
            self._done.synchronize()
        return self._host.numpy()           # the array keeps the pinned block alive


def download(t: torch.Tensor) -> Download:
    return Download(t)


# ---- PIL images on top of the pinned blocks (the PIL end of the batched drivers) ----------------------------------
# Image.fromarray costs 90 us for a 375 x 500 RGB frame — a zero fill of the new image plus a per-pixel repack of packed
# RGB into Pillow's 4-byte RGBX storage — which was 65 % of the batched driver's wall time once the noise draw had moved
# to the device.  Pillow can instead MAP a buffer that already has its storage layout: the device expands the results to
# RGBX (one more byte per pixel over PCIe), Image.frombuffer("RGBX", ...) wraps a frame of the pinned block without
# copying, and ImagingCore.setmode relabels it RGB in place (the call Image.putalpha uses).  The image is flagged
# read-only, so Pillow copies it before any in-place change; it keeps its block alive.  Pinned memory held this way is
# bounded: beyond ZERO_COPY_BUDGET bytes of live images the drivers fall back to Image.fromarray.
ZERO_COPY_BUDGET = int(os.environ.get("IMGXF_PIL_ZERO_COPY_BYTES", str(4 << 30)))
_zc_live = 0
_zc_ok = None


def _zero_copy_probe() -> bool:
    """Does this Pillow map RGBX buffers and relabel them RGB in place?  (Checked once, on a 2 x 3 image.)"""
    try:
        from PIL import Image
        a = np.arange(24, dtype=np.uint8).reshape(2, 3, 4).copy()
        im = Image.frombuffer("RGBX", (3, 2), a, "raw", "RGBX", 0, 1)
        im.im.setmode("RGB")
        im._mode = im.im.mode
        ok = im.mode == "RGB" and np.array_equal(np.asarray(im), a[..., :3])
        a[0, 0, 0] = 200
        return bool(ok and np.asarray(im)[0, 0, 0] == 200 and im.readonly)
    except Exception:
        return False


def zero_copy_reserve(nbytes: int) -> bool:
    """Room for `nbytes` more of pinned memory under live PIL images?  Reserves them if so."""
    global _zc_live, _zc_ok
    if _zc_ok is None:
        _zc_ok = ZERO_COPY_BUDGET > 0 and _zero_copy_probe()
    if not _zc_ok or _zc_live + nbytes > ZERO_COPY_BUDGET:
        return False
    _zc_live += nbytes
    return True


def _zero_copy_release(nbytes: int) -> None:
    global _zc_live
    _zc_live -= nbytes


def image_from_rgbx(frame: np.ndarray):
    """[H, W, 4] uint8 RGBX view of a pinned block (bytes reserved with zero_copy_reserve) -> mode-"RGB" PIL image that
    shares it."""
    from PIL import Image
    h, w = frame.shape[:2]
    im = Image.frombuffer("RGBX", (w, h), frame, "raw", "RGBX", 0, 1)
    im.im.setmode("RGB")
    im._mode = im.im.mode
    weakref.finalize(im, _zero_copy_release, frame.nbytes)
    return im
