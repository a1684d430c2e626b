"""Device JPEG writer throughput: n 4K (or H W) frames → n files; per-kernel split comes from rocprofv3.
usage: python tools/bench_jpeg.py [frames] [H W] ; env KIND=photo|noise"""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from imagetransformations_amd import jpeg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2160, 3840)
KIND = os.environ.get("KIND", "photo")
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
if KIND == "noise":
    frames = torch.randint(0, 256, (N, H, W, 3), dtype=torch.uint8, device=dev, generator=g)
else:   # smooth gradients + mild noise + a few edges: compresses like a photograph (~0.2-0.4 bytes per pixel)
    yy = torch.arange(H, device=dev)[None, :, None, None].float()
    xx = torch.arange(W, device=dev)[None, None, :, None].float()
    ph = torch.arange(N, device=dev)[:, None, None, None].float()
    ch = torch.arange(3, device=dev)[None, None, None, :].float()
    base = 128 + 60 * torch.sin(xx / (90 + 20 * ch) + ph) + 50 * torch.cos(yy / (70 + 10 * ch) + 0.5 * ph) + 30 * ((xx // 256 + yy // 256) % 2)
    base = base + 6 * torch.randn((N, H, W, 3), device=dev, generator=g)
    frames = base.clamp(0, 255).to(torch.uint8)
def timed(fn, steps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps
files, sizes = jpeg.encode_device(frames)
tot = int(sizes.sum().item())
ms = timed(lambda: jpeg.encode_device(frames))
px = N * H * W
print(f"{KIND}: {N} x {H}x{W}: {ms:.3f} ms per batch (device tensors in, files out in HBM)  {px / ms / 1e6:.1f} Gpix/s  "
      f"{N / ms * 1e3:.0f} files/s  {tot / px:.3f} bytes/px", flush=True)
out = jpeg.encode(frames)                                   # warm: pinned staging block allocated
t0 = time.time(); out = jpeg.encode(frames); t1 = time.time()
print(f"  with the copy of the files to the host: {(t1 - t0) * 1e3:.1f} ms  {px / (t1 - t0) / 1e9:.2f} Gpix/s", flush=True)
v = jpeg.encode_views(frames); t0 = time.time(); v = jpeg.encode_views(frames); t1 = time.time()
print(f"  ... as memoryviews of the pinned block (one D2H, no per-file copy): {(t1 - t0) * 1e3:.1f} ms", flush=True)
from PIL import Image
a = frames[0].cpu().numpy()
t0 = time.time()
for _ in range(3):
    buf = io.BytesIO(); Image.fromarray(a).save(buf, "JPEG")
t1 = time.time()
print(f"  Pillow (libjpeg-turbo, one core): {(t1 - t0) / 3 * 1e3:.1f} ms per frame  {H * W * 3 / (t1 - t0) / 1e9:.3f} Gpix/s; "
      f"equal: {buf.getvalue() == out[0]}", flush=True)
from concurrent.futures import ThreadPoolExecutor
ncores = min(16, os.cpu_count() or 1)
arrs = [frames[i % N].cpu().numpy() for i in range(2 * ncores)]
def enc(a):
    b = io.BytesIO(); Image.fromarray(a).save(b, "JPEG"); return len(b.getvalue())
with ThreadPoolExecutor(ncores) as pool:
    list(pool.map(enc, arrs[:ncores]))
    t0 = time.time(); list(pool.map(enc, arrs)); t1 = time.time()
print(f"  Pillow, {ncores} threads (the codec releases the GIL): {len(arrs) / (t1 - t0):.0f} files/s  "
      f"{len(arrs) * H * W / (t1 - t0) / 1e9:.2f} Gpix/s", flush=True)
