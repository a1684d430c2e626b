"""Device JPEG reader against Pillow (libjpeg-turbo) on the host: files/s and Mpix/s for a batch of photo-like files.
usage: python tools/bench_jpeg_decode.py [n_files] [height] [width]   (RESTART=rows adds restart markers every n MCU rows)"""
import io, os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from PIL import Image
from imagetransformations_amd import jpeg_decode

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
h = int(sys.argv[2]) if len(sys.argv) > 2 else 375
w = int(sys.argv[3]) if len(sys.argv) > 3 else 500
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:h, 0:w]
files = []
for i in range(n):
    base = 128 + 60 * np.sin(xx / (17.0 + i % 5)) + 50 * np.cos(yy / 29.0)
    img = np.clip(base[..., None] + 10 * (i % 7) + rng.normal(0, 6, (h, w, 3)), 0, 255).astype(np.uint8)
    buf = io.BytesIO()
    kw = dict(restart_marker_rows=int(os.environ["RESTART"])) if os.environ.get("RESTART") else {}
    Image.fromarray(img).save(buf, "JPEG", quality=75, **kw)
    files.append(buf.getvalue())
mb = sum(len(f) for f in files) / 1e6
print(f"{n} files {h}x{w}, {mb:.1f} MB of JPEG ({mb * 1e6 / (n * h * w):.3f} bytes/px)")
jpeg_decode.decode(files[:4]); torch.cuda.synchronize()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    frames = jpeg_decode.decode(files)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"device reader (host parse + upload + kernels): {dt * 1e3:8.1f} ms  {n / dt:9.1f} files/s  {n * h * w / dt / 1e6:9.1f} Mpix/s")
jpeg_decode.decode(files, profile=True)
print("  stages: " + ", ".join(f"{k} {v * 1e3:.1f} ms" for k, v in jpeg_decode.LAST_PROFILE.items()))
def pil(f): return np.asarray(Image.open(io.BytesIO(f)).convert("RGB"))
t0 = time.perf_counter(); ref = [pil(f) for f in files[:min(n, 64)]]; dt1 = (time.perf_counter() - t0) / min(n, 64)
print(f"Pillow, one core: {dt1 * 1e3:.2f} ms per file  {1 / dt1:9.1f} files/s")
for workers in (8, 16):
    with ThreadPoolExecutor(workers) as pool:
        t0 = time.perf_counter(); out = list(pool.map(pil, files)); dt = time.perf_counter() - t0
    print(f"Pillow, {workers} threads: {dt * 1e3:8.1f} ms  {n / dt:9.1f} files/s")
print("equal:", all(np.array_equal(frames[i].cpu().numpy(), ref[i]) for i in range(len(ref))))
