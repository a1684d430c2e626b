"""Interleaved A/B: apply_scale's resize + centre crop (1.1x Lanczos) on the i8 matrix cores
(resample_mfma.inc) vs the two-pass vector kernels, one process.
usage: python tools/ab_lanczos.py [frames] [rounds] [H W]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetransformations_amd import ops
from imagetransformations_amd.transformation import _scale_t
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H, W = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (2160, 3840)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
frames = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device=dev, generator=g)
def run(scale, no_mfma, iters=6):
    if no_mfma: os.environ["IMGXF_RESAMPLE_NO_MFMA"] = "1"
    else: os.environ.pop("IMGXF_RESAMPLE_NO_MFMA", None)
    __import__("imagetransformations_amd")._ffi.reload_knobs()   # the library caches its knobs
    call = lambda: _scale_t(frames, scale)      # > 1: resize + centre crop; < 1: resize into a black canvas
    call(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): call()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters, W * H
shapes = os.environ.get("AB_SHAPES")
if shapes:       # waves per workgroup "4;8" x AB_OCS chunk lengths
    for shape in shapes.split(";"):
        os.environ["IMGXF_RESAMPLE_MFMA_WAVES"] = shape
        for oc in os.environ.get("AB_OCS", "16").split(","):
            os.environ["IMGXF_RESAMPLE_MFMA_OC"] = oc
            ops._plans.clear()
            ts = [statistics.median([run(sc, False)[0] for _ in range(ROUNDS)]) for sc in (1.1, 1.3, 0.9)]
            print(f"waves={shape:2s} OC={oc:3s}: 1.1x {ts[0]:6.3f}  1.3x {ts[1]:6.3f}  0.9x {ts[2]:6.3f} ms", flush=True)
    sys.exit(0)
ocs = os.environ.get("AB_OCS")
if ocs:
    for oc in ocs.split(","):
        os.environ["IMGXF_RESAMPLE_MFMA_OC"] = oc
        ops._plans.clear()
        t = statistics.median([run(1.1, False)[0] for _ in range(ROUNDS)])
        print(f"1.1x  OC={oc:3s}: {t:7.3f} ms", flush=True)
    sys.exit(0)
for scale in [float(v) for v in os.environ.get("AB_SCALES", "1.1,1.3,1.5,0.9,0.5").split(",")]:
    res = {"two-pass": [], "fused": []}
    for r in range(ROUNDS):
        t, opx = run(scale, True); res["two-pass"].append(t)
        t, opx = run(scale, False); res["fused"].append(t)
    v, m = statistics.median(res["two-pass"]), statistics.median(res["fused"])
    io = 3 * F * (H * W + opx)
    print(f"scale {scale:3.1f}  two-pass {v:7.3f} ms   fused {m:7.3f} ms ({io/m/1e6/8000*100:5.1f}% of 8 TB/s in+out)   x{v/m:.2f}", flush=True)
