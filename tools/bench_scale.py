import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from imagetransformations_amd import ops, transformation as T
dev = torch.device("cuda:0")
g0 = torch.Generator(device=dev); g0.manual_seed(1)
sub = torch.randint(0, 256, (16, 2160, 3840, 3), dtype=torch.uint8, device=dev, generator=g0)
def timeit(fn, it=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
for sc in (0.9, 1.1, 1.2000000000000002, 1.3):
    nw, nh = int(3840 * sc), int(2160 * sc)
    fused = timeit(lambda: T._scale_t(sub, sc))
    if sc > 1:
        l, tp = (nw - 3840) // 2, (nh - 2160) // 2
        two = timeit(lambda: ops.crop(ops.resize_lanczos(sub, (nw, nh)), (l, tp, l + 3840, tp + 2160)))
    else:
        two = float("nan")
    print(f"apply_scale {sc:.1f} (16 4K frames): {fused:.3f} ms   resize then crop: {two:.3f} ms", flush=True)
