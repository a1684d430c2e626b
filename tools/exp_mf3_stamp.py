"""Per-phase cycle sums of sepconv_mfma3_rgb_kernel's steady-state steps (development aid).  Needs a library built with
-DIMGXF_MF3_STAMP=1 (tools/build_variant.sh stamp sepconv_c3.hip -DIMGXF_MF3_STAMP=1): the kernel then writes, per wave,
the s_memtime sums of {barrier wait, P1, P2, P3} and the step count into the fp32 side output.
    python tools/exp_mf3_stamp.py _exp/libimgxf_stamp.so [ksize] [frames]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from imagetransformations_amd import _ffi  # noqa: E402

lib = C.CDLL(sys.argv[1])
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
F = int(sys.argv[3]) if len(sys.argv) > 3 else 64
fn = lib.imgxf_gaussian_u8
fn.restype, fn.argtypes = C.c_int, _ffi.SIGNATURES["imgxf_gaussian_u8"]
H, W = 2160, 3840
dev = torch.device("cuda:0")
src = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device=dev)
out = torch.empty_like(src)
side = torch.zeros((F, H, W, 3), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    rc = fn(_ffi.vp(_ffi.view_of(src)), _ffi.vp(_ffi.view_of(out)), k, k / 6.0, _ffi.vp(_ffi.view_of(side)), st)
    assert rc == 0
torch.cuda.synchronize()
nwg = (W * 3 // 128) * F
t = side.view(-1).view(torch.int64)[: nwg * 4 * 8].view(nwg * 4, 8).cpu().double()
t = t[t[:, 4] > 0]
steps = t[:, 4].sum().item()
names = ["barrier", "P1 H | convert", "P2 V(prev) | split", "P3 V(cur) | flush"]
tot = 0.0
for i, nm in enumerate(names):
    v = t[:, i].sum().item() / steps
    tot += v
    print(f"{nm:22s} {v:8.1f} cycles per step and wave")
print(f"per wave: set-up {t[:, 5].mean().item():9.0f}   loop {t[:, 6].mean().item():9.0f} cycles (max {t[:, 6].max().item():.0f}, min {t[:, 6].min().item():.0f})")
per = side.view(-1).view(torch.int64)[nwg * 32: nwg * 32 + nwg * 4 * 80 * 4].view(nwg * 4, 80, 4).cpu().double()
print("ticks per step by step index (mean over waves): barrier / P1 / P2 / P3 / step")
m = per.mean(0)
for i in range(0, 74, 1):
    print(f"  step {i:2d}: " + " ".join(f"{v:7.1f}" for v in m[i].tolist()) + f"   {m[i].sum().item():8.1f}")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(5):
    fn(_ffi.vp(_ffi.view_of(src)), _ffi.vp(_ffi.view_of(out)), k, k / 6.0, _ffi.vp(_ffi.view_of(side)), st)
ev[1].record(); torch.cuda.synchronize()
print(f"launch time with stamps {ev[0].elapsed_time(ev[1]) / 5:.3f} ms")
print(f"{'step':22s} {tot:8.1f}   ({int(steps)} steady-state steps over {t.shape[0]} waves; s_memtime ticks = shader cycles)")
