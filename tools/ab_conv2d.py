"""Time cv2.filter2D-style convolutions (TransformationPool.motion_blur rows and dense boxes) on 4K frames.
usage: python tools/ab_conv2d.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from imagetransformations_amd import ops
F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H, W = 2160, 3840
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
frames = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device=dev, generator=g)
def motion(size):
    k = np.zeros((size, size)); k[(size - 1) // 2, :] = 1.0 / size
    return k.tolist()
def box(size): return (np.ones((size, size)) / size ** 2).tolist()
def run(kernel, iters=4):
    call = lambda: ops.conv2d(frames, kernel)
    call(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): call()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
px = F * H * W
for name, k in [("motion 5", motion(5)), ("motion 7", motion(7)), ("motion 9", motion(9)), ("motion 11", motion(11)),
                ("box 3x3", box(3)), ("box 5x5", box(5)), ("box 9x9", box(9))]:
    t = run(k)
    print(f"{name:10s} {t:8.3f} ms  {6 * px / t / 1e6 / 8000 * 100:5.1f}% of 8 TB/s", flush=True)
