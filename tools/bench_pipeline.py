"""The reference's whole loop on a directory of JPEG files — load (transformation.py:73-89), eight transformations per
image (:92-157), save (:159-162) — through io_pipeline.run_directory, with each on-disk step on the CPU (Pillow, as the
reference) or on the GPU, and the noise draw from NumPy or from the device generator.
usage: python tools/bench_pipeline.py [n_images] [height] [width]"""
import os, random, shutil, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from PIL import Image
from imagetransformations_amd import io_pipeline as IO, transformation as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
h = int(sys.argv[2]) if len(sys.argv) > 2 else 375
w = int(sys.argv[3]) if len(sys.argv) > 3 else 500
src = tempfile.mkdtemp(prefix="imgxf_pipe_in_")
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:h, 0:w]
for i in range(n):
    base = 128 + 60 * np.sin(xx / (17.0 + i % 5)) + 50 * np.cos(yy / 29.0)
    img = np.clip(base[..., None] + 10 * (i % 7) + rng.normal(0, 6, (h, w, 3)), 0, 255).astype(np.uint8)
    Image.fromarray(img).save(os.path.join(src, f"img_{i:05d}.JPEG"), quality=90)
print(f"{n} files of {h}x{w} in {src}")
try:
    for name, kw, noise in (("Pillow decode, Pillow encode, NumPy noise (round 2 default)", dict(decoder="pillow", encoder="pillow"), "numpy"),
                            ("Pillow decode, device encode, NumPy noise", dict(decoder="pillow", encoder="device"), "numpy"),
                            ("device decode, device encode, NumPy noise", dict(decoder="device", encoder="device"), "numpy"),
                            ("device decode, device encode, device noise", dict(decoder="device", encoder="device"), "device"),
                            ("Pillow decode, Pillow encode, device noise", dict(decoder="pillow", encoder="pillow"), "device")):
        if os.environ.get("ONLY") and os.environ["ONLY"] not in name:
            continue
        dst = tempfile.mkdtemp(prefix="imgxf_pipe_out_")
        try:
            T.NOISE_RNG = noise
            random.seed(0); np.random.seed(0)
            # every configuration runs twice and the second run is reported: the first one pays its own warm-up (kernels and
            # torch ops not launched before, pinned blocks of its sizes, the writer's and reader's tables)
            IO.run_directory(src, dst, chunk_images=256, workers=16, **kw)
            shutil.rmtree(dst); os.makedirs(dst)
            random.seed(0); np.random.seed(0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            files = IO.run_directory(src, dst, chunk_images=256, workers=16, **kw)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f"{name:62s} {dt:7.2f} s  {n / dt:8.1f} images/s  {files / dt:9.1f} files/s", flush=True)
        finally:
            shutil.rmtree(dst, ignore_errors=True)
finally:
    T.NOISE_RNG = "numpy"
    shutil.rmtree(src, ignore_errors=True)
