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
