"""Experiment: the 5x5 Gaussian's launch time in several fresh processes on one box (run it in a shell loop): is the
1.10 vs 1.25 ms spread a per-process state?  Prints the buffers' virtual addresses beside the time.
usage: python tools/exp_process.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetransformations_amd import _ffi
F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H, W = 2160, 3840
dev = torch.device("cuda:0")
cur = torch.cuda.current_stream()
s = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device=dev)
d = torch.empty_like(s)
vs, vd = _ffi.view_of(s), _ffi.view_of(d)
def go(): _ffi.call("imgxf_gaussian_u8", _ffi.vp(vs), _ffi.vp(vd), 5, 5.0 / 6.0, None, cur.cuda_stream)
for _ in range(10): go()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(cur)
    for _ in range(40): go()
    b.record(cur); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 40)
c0 = torch.cuda.Event(enable_timing=True); c1 = torch.cuda.Event(enable_timing=True)
c0.record(cur)
for _ in range(20): d.copy_(s)
c1.record(cur); torch.cuda.synchronize()
print(f"pid {os.getpid()}  src {s.data_ptr():#x} dst {d.data_ptr():#x}  gaussian {min(ts):.4f}..{max(ts):.4f} ms  "
      f"frac {6370099200 * F / 128 / (min(ts) * 1e-3) / 8e12:.3f}  copy {c0.elapsed_time(c1) / 20:.4f} ms", flush=True)
