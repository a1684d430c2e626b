"""CPU baseline leg of bench.py (TEST / MEASUREMENT INFRASTRUCTURE ONLY — never imported by the product).

    python -m oracle.cpu_baseline_run --threads N --budget SECONDS

Runs the C port of the oracle (oracle/c/imgxf_oracle.c, OpenMP) on whole 3840x2160 RGB frames through
the headline step — 5x5 Gaussian + rotate 30 deg / 1.5x bilinear — for about `budget` seconds with N
threads and prints one JSON line.  bench.py starts it as a CHILD PROCESS with OMP_NUM_THREADS=N,
OMP_PROC_BIND=close and OMP_PLACES=cores in the environment (libgomp reads them when it is loaded, which
in bench.py's own process happened long ago with torch), once per point of its thread sweep.
Output / intermediate buffers are allocated once, so the loop measures the arithmetic, not page faults.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H4K, W4K = 2160, 3840


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, required=True)
    ap.add_argument("--budget", type=float, default=8.0)
    args = ap.parse_args()
    from oracle import c_oracle as CO, imgxf_oracle as O
    threads = CO.set_threads(args.threads)
    a = np.random.default_rng(12345).integers(0, 256, (H4K, W4K, 3), dtype=np.uint8)
    m = O.rotate_zoom_matrix(W4K, H4K, 30.0, 1.5)
    g, r, tmp = np.empty_like(a), np.empty_like(a), np.empty(a.size, np.float64)

    def one():
        CO.gaussian_blur(a, 5, 5.0 / 6.0, out=g, tmp=tmp)
        CO.affine(g, (W4K, H4K), m, 1, (0, 0, 0), out=r)

    one()                                    # warm-up: page faults, thread pool
    t0 = time.perf_counter()
    n = 0
    while True:
        one()
        n += 1
        el = time.perf_counter() - t0
        if el > args.budget or n >= 400:
            break
    print(json.dumps({"value": round(n * H4K * W4K / 1e6 / el, 2), "unit": "Mpix/s", "cores": threads, "kind": "port",
                      "frames": n, "seconds": round(el, 2),
                      "omp": {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OMP_PROC_BIND", "OMP_PLACES")}}))


if __name__ == "__main__":
    main()
