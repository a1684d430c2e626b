"""Corrupted scans through the device reader (development aid): byte flips and truncations INSIDE the entropy-coded data of
valid files; every call must return frames or raise ImgxfError / UnsupportedJpeg — never hang or fault.
    python tools/fuzz_jpeg_device.py [cases]"""
import io, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from PIL import Image
from imagetransformations_amd import jpeg_decode, _ffi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(3)
yy, xx = np.mgrid[0:600, 0:800]
img = np.clip((128 + 60 * np.sin(xx / 17.0) + 40 * np.cos(yy / 23.0) + rng.normal(0, 8, (600, 800)))[..., None] + np.zeros(3), 0, 255).astype(np.uint8)
seeds = []
for kw in (dict(quality=85), dict(quality=85, restart_marker_rows=1), dict(quality=60, subsampling=0), dict(quality=92, restart_marker_rows=9)):
    b = io.BytesIO(); Image.fromarray(img).save(b, "JPEG", **kw); seeds.append(b.getvalue())
ok = err = 0
for c in range(n):
    batch = []
    for f in seeds:
        s0, s1 = jpeg_decode.parse(f)["ecs"]
        g = bytearray(f)
        kind = int(rng.integers(0, 3))
        if kind == 0:
            for _ in range(int(rng.integers(1, 30))):
                g[int(rng.integers(s0, s1))] = int(rng.integers(0, 256))
        elif kind == 1:
            g = g[:int(rng.integers(s0, s1))] + b"\xff\xd9"
        else:
            a = int(rng.integers(s0, s1 - 64)); g[a:a + 64] = bytes(rng.integers(0, 256, 64, dtype=np.uint8))
        batch.append(bytes(g))
    try:
        out = jpeg_decode.decode(batch, "cuda"); torch.cuda.synchronize(); ok += 1
    except (_ffi.ImgxfError, jpeg_decode.UnsupportedJpeg):
        err += 1
print("batches", n, "decoded", ok, "reported damaged", err)
good = jpeg_decode.decode(seeds, "cuda")
for t, f in zip(good, seeds):
    assert np.array_equal(t.cpu().numpy(), np.asarray(Image.open(io.BytesIO(f)).convert("RGB")))
print("the undamaged files still decode to Pillow's pixels")
