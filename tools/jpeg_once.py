"""Encode 16 uniform-noise 4K frames CALLS times (bench.py's jpeg_save_q75_4k workload) — run under
`rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU` to count the writer's VALU instructions (tools/collect_jpeg_valu.py)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from imagetransformations_amd import jpeg
CALLS = 3
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(4242)
frames = torch.randint(0, 256, (16, 2160, 3840, 3), dtype=torch.uint8, device=dev, generator=g)
for _ in range(CALLS):
    jpeg.encode_device(frames)
torch.cuda.synchronize()
print("calls", CALLS)
