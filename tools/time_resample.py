"""Time apply_scale's fused resample (1.1x Lanczos + centre crop, 4K RGB) with the library IMGXF_LIBRARY points to
(development aid for experiment builds):  IMGXF_LIBRARY=_exp/libimgxf_x.so python tools/time_resample.py [frames] [scale]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from imagetransformations_amd import ops
F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sc = float(sys.argv[2]) if len(sys.argv) > 2 else 1.1
H, W = 2160, 3840
t = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device="cuda")
nw, nh = int(W * sc), int(H * sc)
box = ((nw - W) // 2, (nh - H) // 2, (nw - W) // 2 + W, (nh - H) // 2 + H) if sc > 1 else (0, 0, nw, nh)
out = ops.resize_crop(t, (nw, nh), box)
ts = []
for r in range(7):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(8): ops.resize_crop(t, (nw, nh), box, out=out)
    e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / 8)
ts.sort()
print(f"{os.environ.get('IMGXF_LIBRARY', 'library'):28s} {F} frames x{sc}: median {ts[3]:.4f} ms   min {ts[0]:.4f}")
