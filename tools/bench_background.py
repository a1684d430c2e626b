import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from imagetransformations_amd import ops
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
g = ops.rgb2l(sub); ed = ops.sobel(g); m = ops.percentile_mask(ed, 70); fg = ops.dilate_cross(m, 3)
bg = ops.new(sub, 2160, 3840, (10, 200, 30))
print("rgb2l", timeit(lambda: ops.rgb2l(sub)))
print("sobel x wrap", timeit(lambda: ops.sobel(g)))
print("percentile_mask (hist+pct+mask)", timeit(lambda: ops.percentile_mask(ed, 70)))
print("channel_histogram(gray)", timeit(lambda: ops.channel_histogram(ed)))
print("dilate", timeit(lambda: ops.dilate_cross(m, 3)))
print("new(bg)", timeit(lambda: ops.new(sub, 2160, 3840, (10, 200, 30))))
print("composite", timeit(lambda: ops.composite(sub, bg, fg)))
print("equalize", timeit(lambda: ops.equalize(sub)))
print("lut(solarize)", timeit(lambda: ops.solarize(sub, 60)))
