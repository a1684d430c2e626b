"""Does the Gaussian's time depend on where the output sits relative to the input?
One 9 GB pool, src at offset 0, dst at 3.19 GB + delta for several deltas (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetransformations_amd import _ffi
dev = torch.device("cuda:0")
F, H, W = 128, 2160, 3840
nb = F * H * W * 3
pool = torch.empty(nb * 2 + (512 << 20), dtype=torch.uint8, device=dev)
src = pool[:nb].view(F, H, W, 3)
g = torch.Generator(device=dev); g.manual_seed(1)
src.copy_(torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device=dev, generator=g))
st = torch.cuda.current_stream().cuda_stream
def run(delta, iters=60):
    dst = pool[nb + delta: nb + delta + nb].view(F, H, W, 3)
    vs, vd = _ffi.view_of(src), _ffi.view_of(dst)
    for _ in range(15): _ffi.call("imgxf_gaussian_u8", _ffi.vp(vs), _ffi.vp(vd), 5, 5/6, None, st)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): _ffi.call("imgxf_gaussian_u8", _ffi.vp(vs), _ffi.vp(vd), 5, 5/6, None, st)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for rep in range(2):
    for delta in (0, 4096, 65536, 1 << 20, 3 << 20, 16 << 20, 100 << 20, 256 << 20):
        print(f"rep {rep} delta {delta:>10d}: {run(delta):.4f} ms", flush=True)
