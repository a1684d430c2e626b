#!/usr/bin/env python3
"""bench.py — headline benchmark of the per-pixel transform hot path on MI355X.

Workload (BASELINE.json `metric`, configs[4] per-GPU share): a resident batch of
independent 3840x2160 RGB uint8 frames; one STEP = 5x5 separable Gaussian (sigma=5/6, the
reference's own ksize rule) followed by rotate-30deg + 1.5x bilinear resample, over every
frame of the batch.  `value` = frames*H*W*steps / time in Mpix/s, whole job (all ranks).

    python bench.py                         # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Frames shard across ranks with no data-path collective (weak scaling: the per-GPU batch is
fixed); each rank synthesises its shard on the device from seed 12345+rank.  Prints ONE JSON
line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
H4K, W4K = 2160, 3840
GAUSS_BYTES_PER_PX = 6.0       # 3 B read + 3 B written per pixel (fused single pass)
AFFINE_BYTES_PER_PX = 4.306    # 3*0.4353 unique source bytes + 3 written
SOBEL_BYTES_PER_PX = 4.0       # RGB in (3) -> u8 magnitude out (1)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=128, help="4K frames resident per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--scatter-gather", action="store_true",
                    help="N > 1 only: also time a root<->ranks scatter + gather of 8 frames per rank "
                         "over RCCL point-to-point (reported separately, never part of `value`)")
    ap.add_argument("--fast-bilinear", action="store_true",
                    help="fp32 interpolation instead of the Pillow-bit-exact fp64 path")
    return ap.parse_args()


def event_ms(fn, iters):
    """Average duration of fn() in ms from HIP events on the current stream."""
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def host_cores() -> int:
    """CPU share of this process: the cgroup quota when there is one, else the affinity mask
    (capped at 16, the per-GPU share of a GPU box, when the mask is the whole host)."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            return max(1, int(int(quota) / int(period)))
    except Exception:
        pass
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return min(n, 16) if n > 32 else n


def pmc_traffic(frames: int):
    """HBM bytes per Gaussian launch from the committed rocprofv3 PMC passes
    (profiles/gaussian_pmc.json, written by tools/collect_traffic.py on the GPU box with the
    same frame count), corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE counts half
    the bytes of a wide coalesced stream on gfx950, WRITE_SIZE is exact; both are in KiB."""
    path = os.path.join(ROOT, "profiles", "gaussian_pmc.json")
    try:
        rec = json.load(open(path))
        if rec.get("frames_per_gpu") != frames:
            return None
        return (2.0 * rec["FETCH_SIZE_KiB"] + rec["WRITE_SIZE_KiB"]) * 1024.0
    except Exception:
        return None


def cpu_baseline():
    """The C port of the oracle (oracle/c/imgxf_oracle.c) on this box's host cores, on a
    bounded sample of the same workload (whole 4K frames through Gaussian + bilinear)."""
    import numpy as np
    from oracle import c_oracle as CO, imgxf_oracle as O
    cores = CO.set_threads(host_cores())
    a = np.random.default_rng(12345).integers(0, 256, (H4K, W4K, 3), dtype=np.uint8)
    m = O.rotate_zoom_matrix(W4K, H4K, 30.0, 1.5)

    def one():
        g = CO.gaussian_blur(a, 5, 5.0 / 6.0)
        return CO.affine(g, (W4K, H4K), m, 1, (0, 0, 0))

    one()                                    # warm-up (page faults, thread pool)
    t0 = time.perf_counter()
    n = 0
    while True:
        one()
        n += 1
        el = time.perf_counter() - t0
        if el > 12.0 or n >= 200:
            break
    return {"value": round(n * H4K * W4K / 1e6 / el, 2), "unit": "Mpix/s", "cores": cores,
            "kind": "port",
            "sample": f"{n} frames of 3840x2160 RGB through 5x5 Gaussian + rotate30/1.5x bilinear "
                      f"(oracle/c, OpenMP, {el:.1f} s)"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device; the HIP path has no CPU fallback")
    # IMGXF_BENCH_BACKEND=gloo + IMGXF_BENCH_SHARED_GPU=1: rehearsal of the N>1 control flow on a
    # one-GPU box (all ranks on device 0, CPU collectives); the driver's runs use nccl (= RCCL)
    backend = os.environ.get("IMGXF_BENCH_BACKEND", "nccl")
    if os.environ.get("IMGXF_BENCH_SHARED_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from imagetransformations_amd import ops   # loads libimgxf.so (raises if missing)

    F = args.frames
    gen = torch.Generator(device=dev)
    gen.manual_seed(12345 + rank)
    frames = torch.randint(0, 256, (F, H4K, W4K, 3), dtype=torch.uint8, device=dev, generator=gen)
    m = ops.rotate_zoom_matrix(W4K, H4K, 30.0, 1.5)
    precise = not args.fast_bilinear

    # pre-allocated outputs: the timed region holds kernels only
    from imagetransformations_amd import _ffi
    blurred = torch.empty_like(frames)
    rotated = torch.empty_like(frames)
    stream = torch.cuda.current_stream().cuda_stream
    vs, vb, vr = _ffi.view_of(frames), _ffi.view_of(blurred), _ffi.view_of(rotated)
    mm, fill = _ffi.f64_array(m), _ffi.u8_array([0, 0, 0])

    def gaussian():
        _ffi.call("imgxf_gaussian_u8", _ffi.vp(vs), _ffi.vp(vb), 5, 5.0 / 6.0, None, stream)

    def rotate():
        _ffi.call("imgxf_affine_u8", _ffi.vp(vb), _ffi.vp(vr), mm, _ffi.FILTER_BILINEAR, fill,
                  1 if precise else 0, None, stream)

    for _ in range(args.warmup):
        gaussian(); rotate()
    torch.cuda.synchronize()
    if dist: dist.barrier()
    torch.cuda.synchronize()

    # events bracket every Gaussian launch so the dominant kernel's duration comes from the
    # timed region itself (same stream the kernels run on)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        gaussian()
        ev[k][1].record()
        rotate()
    torch.cuda.synchronize()
    if dist: dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    gauss_ms = sum(s.elapsed_time(e) for s, e in ev) / args.steps
    px_per_step = F * H4K * W4K
    value = world * px_per_step * args.steps / elapsed / 1e6

    gauss_bytes = GAUSS_BYTES_PER_PX * px_per_step
    achieved = gauss_bytes / (gauss_ms * 1e-3) / 1e9
    result = {
        "metric": "Mpixels/sec, 5x5 Gaussian + bilinear rotate on 4K RGB; % HBM roofline",
        "value": round(value, 1), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/f32",
        "data": "synthetic",
        "config": {"workload": "5x5 separable Gaussian (sigma=5/6, REFLECT_101) + rotate 30deg/1.5x "
                               "bilinear (Pillow semantics) on 3840x2160 RGB uint8 frames",
                   "frames_per_gpu": F, "global_frames": F * world, "height": H4K, "width": W4K,
                   "bilinear": "fp64 bit-exact" if precise else "fp32",
                   "parallelism": f"frames sharded over {world} GPU(s), no data-path collective"},
        "roofline": {"bound": "hbm", "kernel": "sepconv (5x5 Gaussian, 4K RGB)",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(F),
                     "bytes_per_launch": gauss_bytes, "ms_per_launch": round(gauss_ms, 4)},
    }

    if rank == 0 and not args.no_extras:
        extras = {}
        it = 5
        t_g = event_ms(gaussian, it)
        t_r = event_ms(rotate, it)
        extras["gaussian5x5_4k"] = {"Mpix/s": round(px_per_step / t_g / 1e3, 1),
                                    "roofline_frac": round(GAUSS_BYTES_PER_PX * px_per_step / (t_g * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        extras["rotate30_zoom1.5_bilinear_4k"] = {"Mpix/s": round(px_per_step / t_r / 1e3, 1),
                                                  "roofline_frac": round(AFFINE_BYTES_PER_PX * px_per_step / (t_r * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        sub = frames[: min(F, 32)]
        t_s = event_ms(lambda: ops.rgb_sobel_magnitude(sub), it)
        npx = sub.shape[0] * H4K * W4K
        extras["rgb_sobel_magnitude_4k"] = {"Mpix/s": round(npx / t_s / 1e3, 1),
                                            "roofline_frac": round(SOBEL_BYTES_PER_PX * npx / (t_s * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        hd = frames[: min(F, 32), :1080, :1920].contiguous()
        t_h = event_ms(lambda: ops.gaussian_blur(hd, 5, 5.0 / 6.0), it)
        npx = hd.shape[0] * 1080 * 1920
        extras["gaussian5x5_1080p"] = {"Mpix/s": round(npx / t_h / 1e3, 1),
                                       "roofline_frac": round(GAUSS_BYTES_PER_PX * npx / (t_h * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        result["ops"] = extras

    if world > 1 and args.scatter_gather:
        # separate line: root -> ranks scatter and ranks -> root gather of 8 frames
        # per rank over RCCL point-to-point, outside the timed region.  Opt-in: it has only been
        # rehearsed with gloo on one GPU, and a stuck transfer must not cost the headline line
        try:
            from imagetransformations_amd import sharding
            nfr = 8 * world
            root_frames = frames[:8].repeat(world, 1, 1, 1) if rank == 0 else None
            for it in range(2):
                torch.cuda.synchronize(); dist.barrier(); t1 = time.perf_counter()
                local = sharding.scatter_frames(root_frames, nfr, (H4K, W4K, 3), dev)
                back = sharding.gather_frames(local, nfr)
                torch.cuda.synchronize(); dist.barrier(); t2 = time.perf_counter()
            moved = 2 * (nfr - 8) * H4K * W4K * 3          # bytes leaving + re-entering the root
            if rank == 0:
                result["scatter_gather"] = {"GB/s": round(moved / (t2 - t1) / 1e9, 1), "frames": nfr,
                                            "equal": bool(torch.equal(back, root_frames)),
                                            "note": "root<->peers P2P over xGMI, scatter+gather, untimed in `value`"}
        except Exception as exc:                            # noqa: BLE001
            if rank == 0:
                result["scatter_gather"] = {"error": repr(exc)[:200]}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline()
    elif rank == 0:
        result["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(result))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
