import os, sys, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from imagetransformations_amd import _ffi, ops
for (F, H, W) in ((128, 2160, 3840), (512, 1080, 1920), (32, 2160, 3840)):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(3)
    src = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device=dev, generator=g)
    out = torch.empty_like(src)
    vs, vo = _ffi.view_of(src), _ffi.view_of(out)
    m = _ffi.f64_array(ops.rotate_zoom_matrix(W, H, 30.0, 1.5)); fill = _ffi.u8_array([0, 0, 0])
    st = torch.cuda.current_stream().cuda_stream
    for fpb in ("8", "12", "16", "24", "32", "48", "64", "128"):
        os.environ["IMGXF_AFFINE_FPB"] = fpb; _ffi.reload_knobs()
        call = lambda: _ffi.call("imgxf_affine_u8", _ffi.vp(vs), _ffi.vp(vo), m, 1, fill, 1, None, st)
        ts = []
        for _ in range(5):
            call(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(8): call()
            e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / 8)
        print(f"{F} x {H}x{W}  fpb {fpb:4s} {statistics.median(ts):7.4f} ms", flush=True)
    del src, out
