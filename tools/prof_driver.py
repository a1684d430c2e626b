"""The batched in-memory driver (PIL in -> PIL out, 256 images of 375 x 500, eight transformations each) three times in one
process, the third run under cProfile (development aid):  NOISE=device|numpy python tools/prof_driver.py"""
import os, random, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from PIL import Image
from imagetransformations_amd import transformation as T
T.NOISE_RNG = os.environ.get("NOISE", "device")
n,h,w=256,375,500
rng = np.random.default_rng(0); yy, xx = np.mgrid[0:h, 0:w]
base = 128 + 60 * np.sin(xx / 37.0) + 50 * np.cos(yy / 29.0)
imgs = [(Image.fromarray(np.clip(base[..., None] + 10 * (i % 7) + rng.normal(0, 5, (h, w, 3)), 0, 255).astype(np.uint8)), f"img_{i}.jpeg") for i in range(n)]
T.apply_all_transformations_batched(imgs[:8])
for rep in range(3):
    random.seed(0); np.random.seed(0)
    pr=cProfile.Profile()
    torch.cuda.synchronize(); t0=time.perf_counter()
    if rep==2: pr.enable()
    out=T.apply_all_transformations_batched(imgs)
    torch.cuda.synchronize()
    if rep==2: pr.disable()
    print("total", time.perf_counter()-t0, len(out), flush=True)
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
