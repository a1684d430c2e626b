"""Timing-only ablations of affine_bilinear_mf_kernel (development aid): IMGXF_AFFINE_MF_DBG bits
1 = no fp64 hand-back pass, 2 = stage only the first frame of a workgroup, 4 = no gather arithmetic,
8 = no global stores.  Results are WRONG with any bit set; the point is what each phase costs."""
import os, sys, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from imagetransformations_amd import _ffi, ops
F, H, W = int(os.environ.get("FRAMES", "128")), 2160, 3840
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
src = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device=dev, generator=g)
out = torch.empty_like(src)
vs, vo = _ffi.view_of(src), _ffi.view_of(out)
m = _ffi.f64_array(ops.rotate_zoom_matrix(W, H, 30.0, 1.5)); fill = _ffi.u8_array([0, 0, 0])
st = torch.cuda.current_stream().cuda_stream
def run(env, precise=1, iters=8):
    for k in [k for k in os.environ if k.startswith("IMGXF_AFFINE")]: os.environ.pop(k)
    os.environ.update(BASE); os.environ.update(env); _ffi.reload_knobs()
    call = lambda: _ffi.call("imgxf_affine_u8", _ffi.vp(vs), _ffi.vp(vo), m, 1, fill, precise, None, st)
    ts = []
    for _ in range(5):
        call(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters): call()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / iters)
    return statistics.median(ts)
BASE = {"IMGXF_AFFINE_NO_WQ": "1"} if os.environ.get("NO_WQ") else {}
for name, env, pr in [("full precise", {}, 1), ("full fp32", {}, 0), ("mf kernel 32 x 64 (NO_WQ)", {"IMGXF_AFFINE_NO_WQ": "1"}, 1), ("mf kernel 32 x 64 fp32 (NO_WQ)", {"IMGXF_AFFINE_NO_WQ": "1"}, 0),
                      ("mf kernel 64 x 32 (NO_WQ, WIDE)", {"IMGXF_AFFINE_NO_WQ": "1", "IMGXF_AFFINE_MF_WIDE": "1"}, 1),
                      ("no hand-back (1)", {"IMGXF_AFFINE_MF_DBG": "1"}, 1),
                      ("stage first frame only (2)", {"IMGXF_AFFINE_MF_DBG": "2"}, 1),
                      ("no gather (4)", {"IMGXF_AFFINE_MF_DBG": "4"}, 1),
                      ("no stores (8)", {"IMGXF_AFFINE_MF_DBG": "8"}, 1),
                      ("no hand-back, no staging (3)", {"IMGXF_AFFINE_MF_DBG": "3"}, 1),
                      ("no gather, no staging (6)", {"IMGXF_AFFINE_MF_DBG": "6"}, 1),
                      ("no gather, no stores (12)", {"IMGXF_AFFINE_MF_DBG": "12"}, 1),
                      ("only barriers + loop (14)", {"IMGXF_AFFINE_MF_DBG": "14"}, 1),
                      ("gather only: no hand-back/staging/stores (11)", {"IMGXF_AFFINE_MF_DBG": "11"}, 1),
                      ("fp32 gather only (10)", {"IMGXF_AFFINE_MF_DBG": "10"}, 0),
                      ("stores only, 96-byte rows (6)", {"IMGXF_AFFINE_MF_DBG": "6"}, 1),
                      ("stores only, 768 contiguous bytes (22)", {"IMGXF_AFFINE_MF_DBG": "22"}, 1),
                      ("stores only, 192-byte rows (38)", {"IMGXF_AFFINE_MF_DBG": "38"}, 1),
                      ("stores only, 384-byte rows (70)", {"IMGXF_AFFINE_MF_DBG": "70"}, 1),
                      ("full, stores as 768 contiguous (16)", {"IMGXF_AFFINE_MF_DBG": "16"}, 1),
                      ("full, stores as 192-byte rows (32)", {"IMGXF_AFFINE_MF_DBG": "32"}, 1),
                      ("full, stores as 384-byte rows (64)", {"IMGXF_AFFINE_MF_DBG": "64"}, 1),
                      ("staggered start (128)", {"IMGXF_AFFINE_MF_DBG": "128"}, 1),
                      ("staggered start, narrow (128)", {"IMGXF_AFFINE_MF_DBG": "128", "IMGXF_AFFINE_MF_NARROW": "1"}, 1),
                      ("fpb 8", {"IMGXF_AFFINE_FPB": "8"}, 1), ("fpb 32", {"IMGXF_AFFINE_FPB": "32"}, 1),
                      ("3 packed buffers", {"IMGXF_AFFINE_PK3": "1"}, 1)]:
    if os.environ.get("ONLY") and name != os.environ["ONLY"]: continue
    print(f"{name:48s} {run(env, pr):7.4f} ms", flush=True)
