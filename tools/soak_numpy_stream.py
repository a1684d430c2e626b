"""Seeded soak of the device NumPy stream against np.random.normal itself (development aid): random seeds, burn-ins, request
lists (a few samples to several million, odd counts, float32 and float64), numbers and generator state compared.
    python tools/soak_numpy_stream.py [cases] [first_seed]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from imagetransformations_amd import numpy_stream as NS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = patched = 0
for seed in range(s0, s0 + n):
    rng = np.random.default_rng(seed)
    np.random.seed(seed)
    np.random.random_sample(int(rng.integers(0, 2000)))
    if rng.integers(0, 2): np.random.normal(0, 1, int(rng.integers(1, 8)))
    f64 = bool(rng.integers(0, 4) == 0)
    big = seed % 10 == 0
    reqs = [(int(rng.integers(1, 60_000_000 if big and i == 0 else 700_000)), float(rng.uniform(0.01, 100))) for i in range(int(rng.integers(1, 9)))]
    st = np.random.get_state()
    want = [np.random.normal(0, s, c) if f64 else np.random.normal(0, s, c).astype(np.float32) for c, s in reqs]
    after = np.random.get_state()
    np.random.set_state(st)
    got = NS.draw_on_device(reqs, "cuda", f64=f64)
    now = np.random.get_state()
    if f64:       # the doubles may differ in the last bit (they only have to give the same pixel): compare after adding to a pixel grid
        ok = all(np.array_equal(np.clip(np.float32(128.0) + g.cpu().numpy(), 0, 255).astype(np.uint8), np.clip(np.float32(128.0) + w, 0, 255).astype(np.uint8)) for g, w in zip(got, want))
    else:
        ok = all(np.array_equal(g.cpu().numpy(), w) for g, w in zip(got, want))
    ok = ok and now[2] == after[2] and np.array_equal(now[1], after[1]) and now[3] == after[3] and (now[4] == after[4] or not now[3])
    if not ok:
        bad += 1; print("MISMATCH seed", seed, reqs[:2], f64, flush=True)
print("cases", n, "from", s0, "mismatches", bad)
