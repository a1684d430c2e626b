"""Interleaved A/B of the marching Gaussian's launch knobs in ONE process (cdna guide rule 24).
usage: python tools/ab_gauss.py [frames] [rounds] [h] [w]   — prints median / min per variant."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetransformations_amd import _ffi

F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 12
H = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
W = int(sys.argv[4]) if len(sys.argv) > 4 else 3840
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
frames = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device=dev, generator=g)
out = torch.empty_like(frames)
st = torch.cuda.current_stream().cuda_stream
vs, vo = _ffi.view_of(frames), _ffi.view_of(out)
KNOBS = ("IMGXF_MARCH_TAIL", "IMGXF_MARCH_GROUP", "IMGXF_MARCH_SPB", "IMGXF_MARCH_NO_MIXED", "IMGXF_MARCH_RPW", "IMGXF_MARCH_U2")
variants = {
    "r1 (G=1, scalar H, spb4)": {"IMGXF_MARCH_GROUP": "1", "IMGXF_MARCH_NO_MIXED": "1", "IMGXF_MARCH_SPB": "4"},
    "default": {},
    "G=1": {"IMGXF_MARCH_GROUP": "1"},
    "G=2": {"IMGXF_MARCH_GROUP": "2"},
    "G=4": {"IMGXF_MARCH_GROUP": "4"},
    "G=8": {"IMGXF_MARCH_GROUP": "8"},
    "rpw 64": {"IMGXF_MARCH_RPW": "64"},
    "rpw 135": {"IMGXF_MARCH_RPW": "135"},
    "spb 3": {"IMGXF_MARCH_SPB": "3"},
}
extra = os.environ.get("AB_EXTRA")
if extra:      # e.g. AB_EXTRA="rpw180:IMGXF_MARCH_RPW=180"
    for item in extra.split(";"):
        name, kv = item.split(":")
        variants[name] = dict(x.split("=") for x in kv.split(","))

def run(env, iters=10):
    for k in KNOBS: os.environ.pop(k, None)
    os.environ.update(env)
    __import__("imagetransformations_amd")._ffi.reload_knobs()   # the library caches its knobs
    call = lambda: _ffi.call("imgxf_gaussian_u8", _ffi.vp(vs), _ffi.vp(vo), 5, 5 / 6, None, st)
    call(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): call()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

for _ in range(3):
    for env in variants.values(): run(env, 5)
res = {k: [] for k in variants}
for r in range(ROUNDS):
    for name, env in variants.items():
        res[name].append(run(env))
px = F * H * W
for name, v in res.items():
    med, mn = statistics.median(v), min(v)
    print(f"{name:30s} median {med:7.4f} ms  min {mn:7.4f} ms  -> {6 * px / med / 1e6 / 8000 * 100:5.1f}% / {6 * px / mn / 1e6 / 8000 * 100:5.1f}% of 8 TB/s", flush=True)
