"""Interleaved A/B of the batched bilinear rotate+zoom (frames per workgroup) in ONE process.
usage: python tools/ab_affine.py [frames] [rounds]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetransformations_amd import _ffi, ops

F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
H, W = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (2160, 3840)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
frames = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device=dev, generator=g)
out = torch.empty_like(frames)
st = torch.cuda.current_stream().cuda_stream
vs, vo = _ffi.view_of(frames), _ffi.view_of(out)
m = _ffi.f64_array(ops.rotate_zoom_matrix(W, H, 30.0, 1.5)); fill = _ffi.u8_array([0, 0, 0])
variants = {"per-frame kernel (fpb=1)": ("1", 0), "reg staging fpb=16": ("16", 1), "dma fpb=4": ("4", 0), "dma fpb=8": ("8", 0), "dma fpb=16": ("16", 0), "dma fpb=32": ("32", 0)}
if os.environ.get("AB_STRIPS"):  # tile order: row-major ranges vs vertical strips per XCD
    variants = {"dma fpb=16 row-major ranges": ("16", 0, None, None), "dma fpb=16 strips": ("16", 0, None, "1")}
if os.environ.get("AB_PK"):      # two vs three packed-row buffers
    variants = {"dma fpb=16, 3 packed buffers": ("16", 0, "3"), "dma fpb=16, 2 packed buffers": ("16", 0, None),
                "dma fpb=8, 2 packed buffers": ("8", 0, None), "dma fpb=32, 2 packed buffers": ("32", 0, None)}

def run(v, precise, iters=5):
    fpb, nodma = v[0], v[1]
    if len(v) > 2 and v[2]: os.environ["IMGXF_AFFINE_PK3"] = "1"
    else: os.environ.pop("IMGXF_AFFINE_PK3", None)
    if len(v) > 3 and v[3]: os.environ.pop("IMGXF_AFFINE_NO_STRIPS", None)
    elif os.environ.get("AB_STRIPS"): os.environ["IMGXF_AFFINE_NO_STRIPS"] = "1"
    os.environ["IMGXF_AFFINE_FPB"] = fpb
    if nodma: os.environ["IMGXF_AFFINE_NO_DMA"] = "1"
    else: os.environ.pop("IMGXF_AFFINE_NO_DMA", None)
    __import__("imagetransformations_amd")._ffi.reload_knobs()   # the library caches its knobs
    call = lambda: _ffi.call("imgxf_affine_u8", _ffi.vp(vs), _ffi.vp(vo), m, 1, fill, precise, None, st)
    call(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): call()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

for precise in (1, 0):
    res = {k: [] for k in variants}
    for fpb in variants.values(): run(fpb, precise, 2)
    for r in range(ROUNDS):
        for name, fpb in variants.items(): res[name].append(run(fpb, precise))
    px = F * H * W
    for name, v in res.items():
        med = statistics.median(v)
        print(f"{'precise' if precise else 'fp32   '} {name:26s} median {med:7.4f} ms  min {min(v):7.4f} ms -> {4.306 * px / med / 1e6 / 8000 * 100:5.1f}% of 8 TB/s", flush=True)
