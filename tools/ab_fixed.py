"""Interleaved A/B: the fixed-point (OpenCV uint8 evaluation) Gaussian on the i8 matrix cores vs the vector kernels.
usage: python tools/ab_fixed.py [frames] [rounds]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetransformations_amd import ops
F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
H, W = 2160, 3840
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
frames = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device=dev, generator=g)
def run(k, sigma, minr, iters=3):
    os.environ["IMGXF_FX_MFMA_MIN_R"] = str(minr)
    __import__("imagetransformations_amd")._ffi.reload_knobs()   # the library caches its knobs
    call = lambda: ops.gaussian_blur(frames, k, sigma, fixed_point=True)
    call(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): call()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
px = F * H * W
for k, sigma in ((7, 2.5), (9, 3.0), (11, 2.0), (13, 2.0), (15, 2.5), (19, 3.0), (21, 3.5), (25, 4.0), (27, 4.5), (31, 5.0)):
    v = statistics.median([run(k, sigma, 99) for _ in range(ROUNDS)])
    m = statistics.median([run(k, sigma, 2) for _ in range(ROUNDS)])
    print(f"fixed-point k={k:2d}  vector {v:7.3f} ms ({6*px/v/1e6/8000*100:5.1f}%)   i8 matrix cores {m:7.3f} ms ({6*px/m/1e6/8000*100:5.1f}% of 8 TB/s)   x{v/m:.2f}", flush=True)
