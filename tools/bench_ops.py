"""Quick per-op timing on one GPU (development aid; bench.py is the contract benchmark).
usage: python tools/bench_ops.py [gauss|affine|affine_fast|sobel|point|all] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetransformations_amd import ops, _ffi

what = sys.argv[1] if len(sys.argv) > 1 else "all"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 64
H, W = 2160, 3840
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
frames = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device=dev, generator=g)
out = torch.empty_like(frames)
st = torch.cuda.current_stream().cuda_stream
vs, vo = _ffi.view_of(frames), _ffi.view_of(out)

def timeit(fn, iters=10):
    for _ in range(int(os.environ.get("WARM", "1"))): fn()
    torch.cuda.synchronize()
    iters = int(os.environ.get("ITERS", iters))
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

px = F * H * W
def report(name, ms, bpp):
    print(f"{name:34s} {ms:8.3f} ms  {px/ms/1e3:10.0f} Mpix/s  {bpp*px/ms/1e6:8.1f} GB/s  {bpp*px/ms/1e6/8000*100:5.1f}% of 8 TB/s", flush=True)

if what == "gauss5":
    report("gaussian k=5", timeit(lambda: _ffi.call("imgxf_gaussian_u8", _ffi.vp(vs), _ffi.vp(vo), 5, 5/6, None, st), 20), 6.0)
if what in ("gauss", "all"):
    for k, s in ((5, 5/6), (3, 0.5), (7, 1.0), (9, 1.5)):
        report(f"gaussian k={k}", timeit(lambda: _ffi.call("imgxf_gaussian_u8", _ffi.vp(vs), _ffi.vp(vo), k, s, None, st)), 6.0)
if what in ("gauss_big", "all"):
    for k, s in ((13, 2.0), (31, 5.0)):
        report(f"gaussian k={k}", timeit(lambda: _ffi.call("imgxf_gaussian_u8", _ffi.vp(vs), _ffi.vp(vo), k, s, None, st), 3), 6.0)
m = _ffi.f64_array(ops.rotate_zoom_matrix(W, H, 30.0, 1.5)); fill = _ffi.u8_array([0, 0, 0])
if what in ("affine", "all"):
    report("rot30+1.5x bilinear precise", timeit(lambda: _ffi.call("imgxf_affine_u8", _ffi.vp(vs), _ffi.vp(vo), m, 1, fill, 1, None, st), 5), 4.306)
if what in ("affine", "affine_fast", "all"):
    report("rot30+1.5x bilinear fp32", timeit(lambda: _ffi.call("imgxf_affine_u8", _ffi.vp(vs), _ffi.vp(vo), m, 1, fill, 0, None, st), 5), 4.306)
if what in ("affine", "all"):
    m2 = _ffi.f64_array(ops.rotate_matrix(W, H, 30.0))
    report("rot30 nearest", timeit(lambda: _ffi.call("imgxf_affine_u8", _ffi.vp(vs), _ffi.vp(vo), m2, 0, fill, 1, None, st), 5), 5.06)
if what in ("sobel", "all"):
    gray = torch.empty((F, H, W, 1), dtype=torch.uint8, device=dev)
    vg = _ffi.view_of(gray)
    report("rgb->L->sobel magnitude", timeit(lambda: _ffi.call("imgxf_rgb_sobel_mag_u8", _ffi.vp(vs), _ffi.vp(vg), st)), 4.0)
    report("rgb2l", timeit(lambda: _ffi.call("imgxf_rgb2l_u8", _ffi.vp(vs), _ffi.vp(vg), st)), 4.0)
if what in ("point", "all"):
    report("brightness (blend const)", timeit(lambda: _ffi.call("imgxf_blend_u8", None, fill, _ffi.vp(vs), None, _ffi.vp(vo), 1.05, st)), 6.0)
    report("scale_abs", timeit(lambda: _ffi.call("imgxf_scale_abs_u8", _ffi.vp(vs), _ffi.vp(vo), 0.7, 0.0, st)), 6.0)
    report("memcpy d2d (torch copy_)", timeit(lambda: out.copy_(frames)), 6.0)

if what in ("grid", "all"):
    # the remaining transforms of the reference's grid / Pool at the facade's tensor level
    from imagetransformations_amd.transformation import _scale_t, _translation_t
    report("apply_translation (+50, -35)", timeit(lambda: _translation_t(frames, 50, -35), 5), 6.0)
    for size in (5, 9, 11):
        k = [[(1.0 / size if r == (size - 1) // 2 else 0.0) for _ in range(size)] for r in range(size)]
        report(f"motion_blur {size} (filter2D row)", timeit(lambda: ops.conv2d(frames, k), 3), 6.0)
    Fg = min(F, 16)
    for sc in (1.1, 1.3, 0.9):
        ms = timeit(lambda: _scale_t(frames[:Fg], sc), 5)
        print(f"apply_scale x{sc} fused ({Fg} frames)        {ms:8.3f} ms  {Fg*H*W/ms/1e3:10.0f} Mpix/s  {6*Fg*H*W/ms/1e6:8.1f} GB/s  {6*Fg*H*W/ms/1e6/8000*100:5.1f}% of 8 TB/s", flush=True)
    report("flip left-right", timeit(lambda: ops.flip(frames), 5), 6.0)
if what in ("lanczos", "all"):
    Fs = min(F, 16)
    sub = frames[:Fs]
    for sc in (1.1, 0.9, 1.5):
        nw, nh = int(W * sc), int(H * sc)
        ops.resize_lanczos(sub, (nw, nh))          # plan creation outside the timing
        ms = timeit(lambda: ops.resize_lanczos(sub, (nw, nh)), 5)
        opx = Fs * nw * nh
        print(f"lanczos x{sc:<4} ({Fs} frames)          {ms:8.3f} ms  {Fs*H*W/ms/1e3:10.0f} Mpix/s(in)  {(3*Fs*H*W + 3*opx)/ms/1e6:8.1f} GB/s(in+out)", flush=True)
if what in ("misc", "all"):
    Fs = min(F, 16)
    sub = frames[:Fs]
    ms = timeit(lambda: ops.affine(sub, (1, 0.3, -648, 0, 1, 0), (W + 648, H), ops.BICUBIC, (255, 255, 255)), 3)
    print(f"shear 0.3 bicubic ({Fs} frames)        {ms:8.3f} ms  {Fs*H*W/ms/1e3:10.0f} Mpix/s", flush=True)
    noise = torch.randn(sub.shape, device=dev) * 12.0
    ms = timeit(lambda: ops.add_noise(sub, noise), 5)
    print(f"add_noise ({Fs} frames)                {ms:8.3f} ms  {Fs*H*W/ms/1e3:10.0f} Mpix/s  {18*Fs*H*W/ms/1e6:8.1f} GB/s", flush=True)
    g = ops.rgb2l(sub)
    ms = timeit(lambda: ops.percentile_mask(ops.sobel(g), 70), 5)
    print(f"sobel + percentile mask ({Fs} frames)  {ms:8.3f} ms  {Fs*H*W/ms/1e3:10.0f} Mpix/s", flush=True)
    m = ops.percentile_mask(ops.sobel(g), 70)
    ms = timeit(lambda: ops.dilate_cross(m, 3), 5)
    print(f"dilate_cross 3 ({Fs} frames)           {ms:8.3f} ms  {Fs*H*W/ms/1e3:10.0f} Mpix/s", flush=True)
if what in ("persp", "misc", "all"):
    Fs = min(F, 16)
    sub = frames[:Fs]
    # RandomPerspective(0.2)-like corners on a 4K frame
    import numpy as np
    st_ = [[0, 0], [W - 1, 0], [W - 1, H - 1], [0, H - 1]]
    en_ = [[300, 150], [W - 200, 90], [W - 350, H - 120], [120, H - 200]]
    a_ = np.zeros((8, 8)); b_ = np.array(st_, float).reshape(8)
    for i_, (p1, p2) in enumerate(zip(en_, st_)):
        a_[2 * i_] = [p1[0], p1[1], 1, 0, 0, 0, -p2[0] * p1[0], -p2[0] * p1[1]]
        a_[2 * i_ + 1] = [0, 0, 0, p1[0], p1[1], 1, -p2[1] * p1[0], -p2[1] * p1[1]]
    co = [float(v) for v in np.linalg.solve(a_, b_).astype(np.float32)]
    ms = timeit(lambda: ops.perspective(sub, co), 5)
    print(f"perspective warp ({Fs} frames)         {ms:8.3f} ms  {Fs*H*W/ms/1e3:10.0f} Mpix/s  {6*Fs*H*W/ms/1e6:8.1f} GB/s", flush=True)
if what in ("gaussfx", "all"):
    for k, sg in ((5, 5 / 6), (13, 2.0)):
        ms = timeit(lambda: _ffi.call("imgxf_gaussian_cv_fixed_u8", _ffi.vp(vs), _ffi.vp(vo), k, sg, st))
        report(f"gaussian k={k} fixed-point (cv)", ms, 6)
