"""Seeded soak of the reader's parallel entropy decoder against Pillow (development aid; the fixed cases live in
tests/test_gpu_jpeg_decode.py): python tools/soak_jpeg_parallel.py [seeds] [first_seed]"""
import io, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from PIL import Image
from imagetransformations_amd import jpeg_decode

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
batch, meta = [], []
def flush():
    global bad, batch, meta
    if not batch: return
    got = jpeg_decode.decode(batch, "cuda")
    for g, f, m in zip(got, batch, meta):
        want = np.asarray(Image.open(io.BytesIO(f)).convert("RGB"))
        if not np.array_equal(g.cpu().numpy(), want):
            bad += 1; print("MISMATCH", m, flush=True)
    batch, meta = [], []
for seed in range(s0, s0 + n):
    rng = np.random.default_rng(seed)
    h, w = int(rng.integers(120, 1300)), int(rng.integers(120, 1700))
    yy, xx = np.mgrid[0:h, 0:w]
    kind = int(rng.integers(0, 4))
    if kind == 0: img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)                       # noise: long codes, dense scans
    elif kind == 1: img = np.full((h, w, 3), int(rng.integers(0, 256)), np.uint8)              # flat: EOB-only blocks, tiny scans
    else:
        base = 128 + 70 * np.sin(xx / rng.uniform(5, 60)) + 50 * np.cos(yy / rng.uniform(5, 90))
        img = np.clip(base[..., None] + rng.normal(0, rng.uniform(0, 25), (h, w, 3)), 0, 255).astype(np.uint8)
    kw = dict(quality=int(rng.integers(5, 100)), subsampling=int(rng.integers(0, 3)), optimize=bool(rng.integers(0, 2)))
    r = int(rng.integers(0, 4))
    if r == 1: kw["restart_marker_rows"] = int(rng.integers(1, 40))
    if r == 2: kw["restart_marker_blocks"] = int(rng.integers(1, 3000))
    gray = rng.integers(0, 6) == 0
    buf = io.BytesIO()
    try:
        (Image.fromarray(img).convert("L") if gray else Image.fromarray(img)).save(buf, "JPEG", **kw)
    except OSError:
        continue                                              # Pillow's encoder gives up on some optimize / restart combinations
    batch.append(buf.getvalue()); meta.append((seed, h, w, kind, kw, gray))
    if len(batch) == 16: flush()
flush()
print("seeds", n, "from", s0, "mismatches", bad)
