"""Why is the 5x5 Gaussian faster inside bench.py's step loop (alternating with the bilinear kernel) than back to
back?  Per-launch HIP-event times of the SAME launch on the same 128 4K frames under five neighbours:
back to back / alternating with the bilinear kernel / with an idle gap on the host / with a compute-only GEMM /
with a plain device copy.  If an idle GPU between launches helps as much as the bilinear kernel does, the effect is a
duty-cycle (power / clock) one and not a cache or ordering one."""
import os, sys, statistics, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from imagetransformations_amd import _ffi, ops
F, H, W = 128, 2160, 3840
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
src = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device=dev, generator=g)
blur, rot = torch.empty_like(src), torch.empty_like(src)
vs, vb, vr = _ffi.view_of(src), _ffi.view_of(blur), _ffi.view_of(rot)
m = _ffi.f64_array(ops.rotate_zoom_matrix(W, H, 30.0, 1.5)); fill = _ffi.u8_array([0, 0, 0])
st = torch.cuda.current_stream().cuda_stream
A = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16); B = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
gauss = lambda: _ffi.call("imgxf_gaussian_u8", _ffi.vp(vs), _ffi.vp(vb), 5, 5.0 / 6.0, None, st)
others = {
    "back to back": lambda: None,
    "bilinear rotate between": lambda: _ffi.call("imgxf_affine_u8", _ffi.vp(vb), _ffi.vp(vr), m, 1, fill, 1, None, st),
    "host idle 2 ms between": "sleep",
    "bf16 GEMM 8192^3 between": lambda: torch.mm(A, B),
    "device copy of the frames between": lambda: rot.copy_(src),
}
for rep in range(2):
    for name, other in others.items():
        for _ in range(4): gauss(); (other() if callable(other) else None)
        torch.cuda.synchronize()
        ts = []
        for _ in range(24):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); gauss(); e.record()
            if other == "sleep":
                torch.cuda.synchronize(); time.sleep(0.002)
            else:
                other()
            ts.append((s, e))
        torch.cuda.synchronize()
        v = sorted(a.elapsed_time(b) for a, b in ts)
        med = v[len(v) // 2]
        print(f"{name:36s} median {med:7.4f} ms  min {v[0]:7.4f}  max {v[-1]:7.4f}   {6.0 * F * H * W / med / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
