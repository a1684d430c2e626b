"""A few launches of apply_rotation's kernel (NEAREST rotate by 22.5 degrees, 128 4K frames) for rocprofv3 --pmc passes
(development aid):  rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/nearest_once.py [frames] [angle]"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from imagetransformations_amd import ops  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
angle = float(sys.argv[2]) if len(sys.argv) > 2 else 22.5
H, W = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (2160, 3840)
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(1)
src = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device=dev, generator=gen)
out = torch.empty_like(src)
for _ in range(4):
    ops.rotate(src, angle, out=out)
torch.cuda.synchronize()
print("done", int(out[0, H // 2, W // 2, 0]))
