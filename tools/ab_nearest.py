"""Interleaved A/B of the nearest rotation's tile order (row-major ranges vs vertical strips per XCD).
usage: python tools/ab_nearest.py [frames] [rounds]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetransformations_amd import ops
F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
H, W = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (2160, 3840)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
frames = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device=dev, generator=g)
def run(strips, angle, iters=6):
    if strips: os.environ.pop("IMGXF_AFFINE_NO_STRIPS", None)
    else: os.environ["IMGXF_AFFINE_NO_STRIPS"] = "1"
    __import__("imagetransformations_amd")._ffi.reload_knobs()   # the library caches its knobs
    call = lambda: ops.rotate(frames, angle, ops.NEAREST, (0, 0, 0))
    call(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): call()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for angle in (30.0, 22.5, 5.0):
    res = {0: [], 1: []}
    for r in range(ROUNDS):
        for k in (0, 1): res[k].append(run(k, angle))
    print(f"rotate {angle:5.1f} nearest  row-major ranges {statistics.median(res[0]):7.4f} ms   strips {statistics.median(res[1]):7.4f} ms", flush=True)
