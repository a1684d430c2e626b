"""Interleaved A/B: Gaussians on the matrix cores (sepconv_mfma.inc) vs the vector-pipe kernels, one process.
usage: python tools/ab_mfma.py [frames] [rounds]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetransformations_amd import _ffi
F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H, W = 2160, 3840
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
frames = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device=dev, generator=g)
out = torch.empty_like(frames)
st = torch.cuda.current_stream().cuda_stream
vs, vo = _ffi.view_of(frames), _ffi.view_of(out)
def run(k, sigma, minr, iters=4):
    os.environ["IMGXF_MFMA_MIN_R"] = str(minr)
    __import__("imagetransformations_amd")._ffi.reload_knobs()   # the library caches its knobs
    call = lambda: _ffi.call("imgxf_gaussian_u8", _ffi.vp(vs), _ffi.vp(vo), k, sigma, None, st)
    call(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): call()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
px = F * H * W
if os.environ.get("AB_SHAPES"):
    for shape in ("1,34", "1,23", "1,17", "1,12", "1,9", "2,12", "2,9", "1,6"):
        os.environ["IMGXF_MFMA_SHAPE"] = shape
        t = statistics.median([run(31, 5.0, 2) for _ in range(3)])
        print(f"k=31 fpw,bpc = {shape:6s}: {t:7.3f} ms", flush=True)
    sys.exit(0)
if os.environ.get("AB_V2"):         # wave-owned columns (sepconv_mfma2) vs LDS-staged tiles (v1)
    bpcs = os.environ.get("AB_BPC", "").split(",") if os.environ.get("AB_BPC") else [None]
    for k, sigma in ((13, 2.0), (17, 2.5), (19, 3.0), (25, 4.0), (31, 5.0)):
        a = []
        os.environ["IMGXF_MFMA_V1"] = "1"
        for r in range(ROUNDS): a.append(run(k, sigma, 2))
        os.environ.pop("IMGXF_MFMA_V1", None)
        line = f"k={k:2d}  v1 {statistics.median(a):7.3f} ms"
        for bpc in bpcs:
            if bpc: os.environ["IMGXF_MFMA2_BPC"] = bpc
            b = [run(k, sigma, 2) for r in range(ROUNDS)]
            line += f"   v2{' bpc=' + bpc if bpc else ''} {statistics.median(b):7.3f} ms"
        os.environ.pop("IMGXF_MFMA2_BPC", None)
        print(line, flush=True)
    sys.exit(0)
if os.environ.get("AB_HREG"):       # H operand tables from LDS vs registers
    for k, sigma in ((13, 2.0), (19, 3.0), (25, 4.0), (31, 5.0)):
        a, b = [], []
        for r in range(ROUNDS):
            os.environ["IMGXF_MFMA_NO_HREG"] = "1"; a.append(run(k, sigma, 2))
            os.environ.pop("IMGXF_MFMA_NO_HREG", None); b.append(run(k, sigma, 2))
        print(f"k={k:2d}  H tables in LDS {statistics.median(a):7.3f} ms   in registers {statistics.median(b):7.3f} ms", flush=True)
    sys.exit(0)
for k, sigma in ((5, 5 / 6), (7, 1.0), (9, 1.5), (13, 2.0), (15, 2.5), (19, 3.0), (21, 3.5), (25, 4.0), (27, 4.5), (31, 5.0)):
    res = {"vector": [], "mfma": []}
    for r in range(ROUNDS):
        res["vector"].append(run(k, sigma, 99)); res["mfma"].append(run(k, sigma, 2))
    v, m = statistics.median(res["vector"]), statistics.median(res["mfma"])
    print(f"k={k:2d}  vector {v:7.3f} ms ({6*px/v/1e6/8000*100:5.1f}%)   mfma {m:7.3f} ms ({6*px/m/1e6/8000*100:5.1f}% of 8 TB/s)   x{v/m:.2f}", flush=True)
