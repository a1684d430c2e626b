"""Experiment: does the 5x5 Gaussian's speed (two modes seen, ~1.10 and ~1.24 ms per 64 4K frames) depend on where
its buffers landed?  Times the same launch on several (src, dst) pairs allocated in one process with different
paddings between them, then on one big arena with dst at several offsets from src.
usage: python tools/exp_alloc.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetransformations_amd import _ffi
F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H, W = 2160, 3840
dev = torch.device("cuda:0")
cur = torch.cuda.current_stream()
n = F * H * W * 3
def view(buf, off):
    return buf[off:off + n].view(F, H, W, 3)
def timed(s, d, steps=20):
    vs, vd = _ffi.view_of(s), _ffi.view_of(d)
    def go(): _ffi.call("imgxf_gaussian_u8", _ffi.vp(vs), _ffi.vp(vd), 5, 5.0 / 6.0, None, cur.cuda_stream)
    for _ in range(5): go()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(cur)
        for _ in range(steps): go()
        b.record(cur); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / steps)
    return min(ts), max(ts)
def tcopy(s, d, steps=20):
    for _ in range(3): d.copy_(s)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(cur)
    for _ in range(steps): d.copy_(s)
    b.record(cur); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps
print("separate allocations (padding allocated between src and dst)", flush=True)
keep = []
for pad in (0, 1 << 20, 3 << 20, 64 << 20, 257 << 20, 1 << 30, 0, 0):
    s = torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev)
    if pad: keep.append(torch.empty(pad, dtype=torch.uint8, device=dev))
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    lo, hi = timed(view(s, 0), view(d, 0))
    print(f"pad {pad >> 20:5d} MiB  src {s.data_ptr():#x} dst {d.data_ptr():#x}  gaussian {lo:.3f}..{hi:.3f} ms   copy {tcopy(s, d):.3f} ms", flush=True)
    keep.append((s, d))
del keep
torch.cuda.empty_cache()
print("one arena, dst offset from the end of src", flush=True)
arena = torch.randint(0, 256, (2 * n + (64 << 20),), dtype=torch.uint8, device=dev)
for off in (0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, 5 << 20, 17 << 20, 32 << 20, 33 << 20):
    lo, hi = timed(view(arena, 0), view(arena, n + off))
    print(f"offset {off:9d}  gaussian {lo:.3f}..{hi:.3f} ms", flush=True)
