"""Experiment: the bench step's two kernels (memory-bound 5x5 Gaussian, VALU-bound bilinear rotate) on ONE stream
back to back vs on TWO streams concurrently.  usage: python tools/exp_overlap.py [frames] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetransformations_amd import ops
F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
H, W = 2160, 3840
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
frames = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device=dev, generator=g)
m = ops.rotate_zoom_matrix(W, H, 30.0, 1.5)
out_g = torch.empty_like(frames); out_r = torch.empty_like(frames)
from imagetransformations_amd import _ffi
vs, vg, vr = _ffi.view_of(frames), _ffi.view_of(out_g), _ffi.view_of(out_r)
mm = _ffi.f64_array(m); fill = _ffi.u8_array([0, 0, 0])
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def gauss(st): _ffi.call("imgxf_gaussian_u8", _ffi.vp(vs), _ffi.vp(vg), 5, 5.0 / 6.0, None, st.cuda_stream)
def rot(st): _ffi.call("imgxf_affine_u8", _ffi.vp(vs), _ffi.vp(vr), mm, 1, fill, 1, None, st.cuda_stream)
def timed(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(torch.cuda.current_stream())
    for _ in range(STEPS): fn()
    b.record(torch.cuda.current_stream()); torch.cuda.synchronize()
    return a.elapsed_time(b) / STEPS
cur = torch.cuda.current_stream()
def serial():
    gauss(cur); rot(cur)
def overlapped():
    e = torch.cuda.Event(); e.record(cur)
    s1.wait_event(e); s2.wait_event(e)
    gauss(s1); rot(s2)
    e1, e2 = torch.cuda.Event(), torch.cuda.Event()
    e1.record(s1); e2.record(s2)
    cur.wait_event(e1); cur.wait_event(e2)
px = 2 * F * H * W
for name, fn in (("one stream", serial), ("two streams", overlapped), ("one stream", serial), ("two streams", overlapped)):
    t = timed(fn)
    print(f"{name:12s} {t:7.3f} ms per step  {px / t / 1e6:8.1f} Gpix/s", flush=True)
