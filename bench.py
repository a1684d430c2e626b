#!/usr/bin/env python3
"""bench.py — headline benchmark of the per-pixel transform hot path on MI355X.

Workload (BASELINE.json `metric`, configs[4] per-GPU share): a resident batch of
independent 3840x2160 RGB uint8 frames; one STEP = 5x5 separable Gaussian (sigma=5/6, the
reference's own ksize rule, /root/reference/transformation.py:239-249) followed by
rotate-30deg + 1.5x bilinear resample (Pillow semantics, configs[3]), over every frame of the
batch.  `value` = frames*H*W*steps / time in Mpix/s, whole job (all ranks).  The same step is
also timed on 1920x1080 frames (north_star: both resolutions) and reported under `resolutions`.

    python bench.py                         # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Frames shard across ranks with no data-path collective (weak scaling: the per-GPU batch is
fixed); each rank synthesises its shard on the device from seed 12345+rank.  Prints ONE JSON
line on rank 0.  For N > 1 a root<->ranks scatter + gather over RCCL point-to-point is measured
in fresh child processes with a timeout (reported separately, never part of `value`).
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import statistics
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
H4K, W4K = 2160, 3840
HHD, WHD = 1080, 1920
GAUSS_BYTES_PER_PX = 6.0       # 3 B read + 3 B written per pixel (fused single pass)
AFFINE_BYTES_PER_PX = 4.306    # 3*0.4353 unique source bytes + 3 written
SOBEL_BYTES_PER_PX = 4.0       # RGB in (3) -> u8 magnitude out (1)
METRIC = "Mpixels/sec, 5x5 Gaussian + bilinear rotate on 4K RGB; % HBM roofline"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=128, help="4K frames resident per GPU (1080p: 4x as many)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-1080p", action="store_true")
    ap.add_argument("--only-1080p", action="store_true",
                    help="PMC passes: run only the 1920x1080 step (4x as many frames) and print a minimal line")
    ap.add_argument("--no-scatter-gather", action="store_true",
                    help="N > 1: skip the separate root<->ranks scatter + gather measurement")
    ap.add_argument("--scatter-gather", action="store_true", help="(default for N > 1; kept for compatibility)")
    ap.add_argument("--sg-timeout", type=float, default=120.0)
    ap.add_argument("--sg-child", action="store_true", help=argparse.SUPPRESS)
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


def pci_bdf(index: int):
    """PCI address ("0000:c1:00.0") of HIP device `index` (hipDeviceGetPCIBusId; torch's properties as fallback)."""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(index)) == 0 and buf.value:
            return buf.value.decode().lower()
    except Exception:
        pass
    try:
        p = torch.cuda.get_device_properties(index)
        return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception:
        return None


class SclkSampler:
    """Samples THIS device's shader clock from sysfs in a background thread while the timed region runs.  The HIP
    ordinal is resolved to its PCI address first (/sys/bus/pci/devices/<bdf>/hwmon/hwmon*/freq1_input, else that
    device's pp_dpm_sclk): on a multi-card host with one visible device the n-th card of a sorted glob is a
    neighbour's idle clock.  None when the device's files cannot be found — never another card's."""

    def __init__(self, index: int):
        self.bdf = pci_bdf(index)
        self.hwmon = self.path = None
        if self.bdf:
            base = f"/sys/bus/pci/devices/{self.bdf}"
            hw = sorted(glob.glob(base + "/hwmon/hwmon*/freq1_input"))
            self.hwmon = hw[0] if hw else None
            dpm = base + "/pp_dpm_sclk"
            self.path = self.hwmon or (dpm if os.path.exists(dpm) else None)
        self.samples, self._stop, self._thr = [], threading.Event(), None

    def _read(self):
        try:
            if self.hwmon:
                return float(open(self.hwmon).read().strip()) / 1e6
            for line in open(self.path).read().splitlines():
                if line.rstrip().endswith("*"):
                    return float(line.split(":")[1].strip().lower().replace("mhz", "").replace("*", "").strip())
        except Exception:
            return None
        return None

    def start(self):
        if self.path is None:
            return
        def loop():
            while not self._stop.is_set():
                v = self._read()
                if v is not None:
                    self.samples.append(v)
                time.sleep(0.004)
        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()

    def stop(self):
        if self._thr is None:
            return None
        self._stop.set()
        self._thr.join()
        if not self.samples:
            return None
        s = sorted(self.samples)
        return {"min": s[0], "median": s[len(s) // 2], "max": s[-1], "samples": len(s), "pci": self.bdf,
                "source": "sysfs hwmon freq1_input" if self.hwmon else "sysfs pp_dpm_sclk"}


def in_kernel_sclk(dev, work, ms: float = 20.0):
    """Shader clock as the kernels see it (MI355X_MICROARCH.md): a one-wave probe on a SIDE stream stamps
    s_memtime (shader cycles) and s_memrealtime (constant 100 MHz) around `ms` milliseconds while `work()` keeps
    the step's kernels running on the main stream; MHz = 100 * d(memtime) / d(memrealtime).  Outside the timed
    region."""
    from imagetransformations_amd import _ffi
    out = torch.zeros(2, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(device=dev)
    for _ in range(3):
        work()
    torch.cuda.synchronize()
    work()
    _ffi.call("imgxf_probe_sclk", out.data_ptr(), int(ms * 1e5), side.cuda_stream)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < ms * 1.5e-3:       # keep the queue full for the probe's whole window
        work()
    torch.cuda.synchronize()
    c, r = (int(v) for v in out.tolist())
    if r <= 0 or c <= 0:
        return None
    return {"mhz": round(100.0 * c / r, 1), "window_ms": round(r / 1e5, 2),
            "how": "d s_memtime / d s_memrealtime x 100 MHz, one wave on a side stream under the step's kernels"}


def pmc_traffic(kernel_key: str, frames: int, section: str = "kernels"):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic_pmc.json,
    written by tools/collect_traffic.py on the GPU box with the same frame count), corrected as
    MI355X_MICROARCH.md prescribes: FETCH_SIZE counts half the bytes of a wide coalesced stream on
    gfx950, WRITE_SIZE is exact; both are in KiB."""
    for name in ("traffic_pmc.json", "gaussian_pmc.json"):
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", name)))
            if rec.get("frames_per_gpu") != frames:
                continue
            k = rec.get(section, {}).get(kernel_key) if "kernels" in rec else (rec if kernel_key == "sepconv" else None)
            if k:
                return (2.0 * k["FETCH_SIZE_KiB"] + k["WRITE_SIZE_KiB"]) * 1024.0
        except Exception:
            pass
    return None


def cpu_baseline_point(threads: int, budget_s: float):
    """One point of the CPU baseline: the C port of the oracle (oracle/c/imgxf_oracle.c, OpenMP) in a CHILD process
    (oracle/cpu_baseline_run.py) with `threads` bound close to cores, on a bounded sample of the same workload."""
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="close", OMP_PLACES="cores")
    try:
        p = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline_run", "--threads", str(threads), "--budget", str(budget_s)],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=budget_s * 6 + 120)
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not lines:
            return {"error": f"rc={p.returncode}: {p.stderr[-200:]}", "cores": threads}
        rec = json.loads(lines[-1])
    except Exception as exc:                                # noqa: BLE001
        return {"error": repr(exc)[:200], "cores": threads}
    rec["sample"] = (f"{rec.pop('frames')} frames of 3840x2160 RGB through 5x5 Gaussian + rotate30/1.5x bilinear "
                     f"(oracle/c, OpenMP close/cores, {rec.pop('seconds'):.1f} s)")
    return rec


def cpu_baselines():
    """`cpu_baseline`: the port on this process's CPU share (cgroup quota / affinity; 16 on a one-GPU box).
    `cpu_baseline_all_cores`: the BEST point of a sweep over 16 / 32 / 64 / 128 / all host threads (each bound
    close to cores) with every point listed — on a box whose cgroup grants 16 CPUs more threads only thrash, and
    the record shows that instead of quoting the oversubscribed figure."""
    share, allc = host_cores(), os.cpu_count() or 1
    base = cpu_baseline_point(share, 8.0)
    points = sorted({n for n in (16, 32, 64, 128, allc) if share < n <= allc})
    sweep = [base] + [cpu_baseline_point(n, 3.0) for n in points]
    good = [p for p in sweep if "value" in p]
    best = max(good, key=lambda p: p["value"]) if good else None
    allrec = None
    if best is not None:
        allrec = dict(best)
        allrec["host_threads"] = allc
        allrec["sweep"] = [{"cores": p.get("cores"), "value": p.get("value"), "error": p.get("error")} for p in sweep]
    return base, allrec


def stats(ms):
    s = sorted(ms)
    return {"median": round(s[len(s) // 2], 4), "min": round(s[0], 4), "max": round(s[-1], 4),
            "mean": round(sum(s) / len(s), 4)}


def timed_step(H, W, F, args, rank, world, dist, dev, backend, precise):
    """W warm-up + K timed steps of Gaussian + rotate/zoom over F resident HxW frames; HIP events
    bracket every launch on the launch stream.  Returns (elapsed max over ranks, per-launch ms)."""
    from imagetransformations_amd import _ffi, ops
    gen = torch.Generator(device=dev)
    gen.manual_seed(12345 + rank)
    frames = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device=dev, generator=gen)
    m = ops.rotate_zoom_matrix(W, H, 30.0, 1.5)
    blurred = torch.empty_like(frames)       # pre-allocated outputs: the timed region holds kernels only
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
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        gaussian()
        ev[k][1].record()
        rotate()
        ev[k][2].record()
    torch.cuda.synchronize()
    if dist: dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    g_ms = [e[0].elapsed_time(e[1]) for e in ev]
    r_ms = [e[1].elapsed_time(e[2]) for e in ev]
    return elapsed, g_ms, r_ms, (frames, gaussian, rotate)


def kernel_entry(name, bytes_per_px, px, ms, traffic):
    st = stats(ms)
    achieved = bytes_per_px * px / (st["mean"] * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_median": round(bytes_per_px * px / (st["median"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "traffic": traffic, "bytes_per_launch": bytes_per_px * px, "ms_per_launch": st["mean"], "ms": st}


def launch_stats(fn, iters: int, warm: int = 3):
    """Per-launch HIP-event times (ms) of fn() on the current stream: the same instrument as the headline step."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for k in range(iters):
        fn()
        ev[k + 1].record()
    torch.cuda.synchronize()
    return [ev[k].elapsed_time(ev[k + 1]) for k in range(iters)]


def ops_extras(F: int, dev, sclk_kernel):
    """The other transforms of the reference's grid (transformation.py:95-105) and benchmark configs[2], each through the
    C-ABI into a PRE-ALLOCATED output on the SAME frame count as the headline step (F 4K frames, 10 launches, per-launch
    events) so that the fractions are comparable with `roofline_kernels`."""
    from imagetransformations_amd import ops, jpeg
    extras = {}
    gen = torch.Generator(device=dev); gen.manual_seed(4242)
    sub = torch.randint(0, 256, (F, H4K, W4K, 3), dtype=torch.uint8, device=dev, generator=gen)
    out = torch.empty_like(sub)
    gray = torch.empty((F, H4K, W4K, 1), dtype=torch.uint8, device=dev)
    npx = F * H4K * W4K

    def entry(fn, bytes_per_px, iters=10, **more):
        st = stats(launch_stats(fn, iters))
        rec = {"Mpix/s": round(npx / st["mean"] / 1e3, 1), "frames": F, "ms": st}
        if bytes_per_px is not None:
            rec["roofline_frac"] = round(bytes_per_px * npx / (st["mean"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            rec["bytes_per_px"] = bytes_per_px
        rec.update(more)
        return rec

    extras["rgb_sobel_magnitude_4k"] = entry(lambda: ops.rgb_sobel_magnitude(sub, out=gray), SOBEL_BYTES_PER_PX)
    extras["gaussian5x5_4k_back_to_back"] = entry(lambda: ops.gaussian_blur(sub, 5, 5.0 / 6.0, out=out), GAUSS_BYTES_PER_PX)
    nw, nh = int(W4K * 1.1), int(H4K * 1.1)                       # apply_scale 1.1x: Lanczos resize + centre crop (:173-196)
    left, top = (nw - W4K) // 2, (nh - H4K) // 2
    extras["apply_scale_1.1_lanczos_crop_4k"] = entry(
        lambda: ops.resize_crop(sub, (nw, nh), (left, top, left + W4K, top + H4K), out=out), 6.0,
        kernel="resample_mfma_kernel (both passes on the i8 matrix cores)")
    m30 = ops.rotate_zoom_matrix(W4K, H4K, 30.0, 1.5)
    extras["rotate30_zoom1.5_bilinear_fp32_mode_4k"] = entry(       # the <= 1e-5 contract without the Pillow-exact guard / redo
        lambda: ops.affine(sub, m30, (W4K, H4K), ops.BILINEAR, (0, 0, 0), precise=False, out=out), AFFINE_BYTES_PER_PX)
    extras["rotate30_zoom1.5_bilinear_precise_back_to_back_4k"] = entry(
        lambda: ops.affine(sub, m30, (W4K, H4K), ops.BILINEAR, (0, 0, 0), precise=True, out=out), AFFINE_BYTES_PER_PX)
    extras["apply_rotation_22.5_nearest_4k"] = entry(lambda: ops.rotate(sub, 22.5, ops.NEAREST, (0, 0, 0), out=out), None)
    extras["gaussian31x31_4k"] = entry(lambda: ops.gaussian_blur(sub, 31, 5.0, out=out), GAUSS_BYTES_PER_PX, iters=5,
                                       kernel="sepconv_mfma2_rgb_kernel (f16 matrix cores)")
    # the save step (transformation.py:161-162): one JPEG file per frame, byte-identical to Pillow's; noise frames are the
    # encoder's worst case (0.6 bytes per pixel of entropy-coded data; photographs: ~0.1).  Its bound is integer VALU
    # issue, so the fraction quoted is VALU issue slots used / available: wave64 VALU instructions (rocprofv3
    # SQ_INSTS_VALU per pixel, profiles/jpeg_valu_pmc.json) x 4 cycles / (1024 SIMDs x in-kernel clock x time)
    sub16 = sub[:16]
    n16 = 16 * H4K * W4K
    st = stats(launch_stats(lambda: jpeg.encode_device(sub16), 5))
    _, jsz = jpeg.encode_device(sub16)
    jbytes = float(jsz.sum().item())
    rec = {"Mpix/s": round(n16 / st["mean"] / 1e3, 1), "files/s": round(16 / st["mean"] * 1e3, 1), "frames": 16, "ms": st,
           "file_bytes_per_px": round(jbytes / n16, 3),
           "kernels": "jpeg_transform + jpeg_lens + scan + jpeg_emit + jpeg_ffcount + scan + jpeg_stuff",
           "bound": "integer VALU issue", "valu_issue_frac": None,
           "hbm_frac_for_reference": round((3.0 * n16 + jbytes) / (st["mean"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "jpeg_valu_pmc.json")))
        if sclk_kernel and pmc.get("valu_insts_per_px"):
            insts = pmc["valu_insts_per_px"] * n16
            rec["valu_issue_frac"] = round(insts * 4.0 / (1024 * sclk_kernel["mhz"] * 1e6 * st["mean"] * 1e-3), 4)
            rec["valu_insts_per_px"] = pmc["valu_insts_per_px"]
    except Exception:
        pass
    extras["jpeg_save_q75_4k"] = rec
    # the load step (transformation.py:83) on the device: 256 photograph-like 375 x 500 files (the reference's ImageNet size,
    # written by the device writer = the files Pillow would write) -> RGB frames; host parsing and the upload of the
    # compressed bytes are inside the time.  The entropy stage is latency bound (one lane per file: DESIGN 3.8)
    try:
        from imagetransformations_amd import jpeg_decode
        yy = torch.arange(375, device=dev)[None, :, None, None].float(); xx = torch.arange(500, device=dev)[None, None, :, None].float()
        ph = torch.arange(256, device=dev)[:, None, None, None].float(); ch = torch.arange(3, device=dev)[None, None, None, :].float()
        small = (128 + 60 * torch.sin(xx / (17 + ch) + ph) + 50 * torch.cos(yy / 29 + 0.3 * ph) +
                 6 * torch.randn((256, 375, 500, 3), device=dev, generator=gen)).clamp(0, 255).to(torch.uint8)
        files = jpeg.encode(small)
        jpeg_decode.decode(files[:8], dev)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            frames = jpeg_decode.decode(files, dev)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        jpeg_decode.decode(files, dev, profile=True)
        t = sorted(ts)[1]
        extras["jpeg_load_375x500"] = {"files/s": round(256 / t, 1), "Mpix/s": round(256 * 375 * 500 / t / 1e6, 1), "files": 256,
                                       "ms": round(t * 1e3, 2), "file_bytes_per_px": round(sum(len(f) for f in files) / (256 * 375 * 500), 3),
                                       "stages_ms": {k: round(v * 1e3, 2) for k, v in jpeg_decode.LAST_PROFILE.items()},
                                       "bound": "latency chain host layout -> upload -> three kernels; entropy decoding runs 256 self-synchronising subsequences per image"}
        del small, frames
    except Exception as exc:                                # noqa: BLE001
        extras["jpeg_load_375x500"] = {"error": repr(exc)[:200]}
    del sub, sub16, out, gray
    torch.cuda.empty_cache()
    return extras


def verify_checksum(rank, world, dist, dev, precise):
    """A fixed-seed batch of 4K frames (the same on every rank) goes through Gaussian + rotate sharded over the ranks
    exactly as the workload is (contiguous blocks); the position-weighted checksum of the sharded results, summed
    over the ranks, must equal the one the root computes for the whole batch alone.  At world size 1 the
    batch is processed once whole and once in two blocks (frames must not depend on their position in a launch)."""
    from imagetransformations_amd import ops, sharding
    n = max(8, 2 * world)
    gen = torch.Generator(device=dev); gen.manual_seed(777)
    full = torch.randint(0, 256, (n, H4K, W4K, 3), dtype=torch.uint8, device=dev, generator=gen)
    m = ops.rotate_zoom_matrix(W4K, H4K, 30.0, 1.5)

    def step(t):
        return ops.affine(ops.gaussian_blur(t, 5, 5.0 / 6.0), m, (W4K, H4K), ops.BILINEAR, (0, 0, 0), precise=precise)

    if world > 1:
        a, b = sharding.shard_range(n, world, rank)
        sharded = sharding.checksum_weighted(step(full[a:b]), offset=a)            # all-reduce over the ranks
        alone = sharding.checksum_weighted(step(full), offset=0, reduce=False) if rank == 0 else None
    else:
        alone = sharding.checksum_weighted(step(full), offset=0, reduce=False)
        h = n // 2
        sharded = (sharding.checksum_weighted(step(full[:h]), offset=0, reduce=False) +
                   sharding.checksum_weighted(step(full[h:]), offset=h, reduce=False))
        sharded = (sharded + 2 ** 63) % 2 ** 64 - 2 ** 63                           # wrap like int64
    del full
    torch.cuda.empty_cache()
    if rank != 0:
        return None, None
    return bool(sharded == alone), {"sharded": sharded, "single": alone, "frames": n,
                                    "what": "sum_j (j+1) sum_p byte[j,p] (p mod 65521 + 1) over Gaussian + rotate of a seed-777 batch"}


def sg_child(args):
    """Fresh-process leg of the scatter/gather measurement (its own process group on MASTER_PORT
    of the environment it was started with): 8 4K frames per rank leave the root and come back."""
    import torch.distributed as dist
    from imagetransformations_amd import sharding
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = 0 if os.environ.get("IMGXF_BENCH_SHARED_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = os.environ.get("IMGXF_BENCH_BACKEND", "nccl")
    dist.init_process_group(backend, device_id=dev) if backend == "nccl" else dist.init_process_group(backend)
    nfr = 8 * world
    gen = torch.Generator(device=dev); gen.manual_seed(999)
    root_frames = torch.randint(0, 256, (nfr, H4K, W4K, 3), dtype=torch.uint8, device=dev, generator=gen) if rank == 0 else None
    host = backend != "nccl"
    if host and root_frames is not None:
        root_frames = root_frames.cpu()
    ddev = torch.device("cpu") if host else dev
    t1 = t2 = 0.0
    back = None
    for _ in range(2):
        torch.cuda.synchronize(); dist.barrier(); t1 = time.perf_counter()
        loc = sharding.scatter_frames(root_frames, nfr, (H4K, W4K, 3), ddev)
        back = sharding.gather_frames(loc, nfr)
        torch.cuda.synchronize(); dist.barrier(); t2 = time.perf_counter()
    if rank == 0:
        moved = 2 * (nfr - 8) * H4K * W4K * 3            # bytes leaving + re-entering the root
        print(json.dumps({"GB/s": round(moved / (t2 - t1) / 1e9, 1), "frames": nfr, "backend": backend,
                          "equal": bool(torch.equal(back, root_frames)),
                          "note": "root<->peers P2P (RCCL send/recv over xGMI), scatter+gather, never part of `value`"}))
    dist.barrier()
    dist.destroy_process_group()


def run_sg_children(args, rank, world, dist, dev, backend):
    """Every rank starts a fresh child (same RANK / WORLD_SIZE, rendezvous on MASTER_PORT + 1) so a
    stuck transfer can be killed after --sg-timeout without costing the headline line."""
    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
    # under torchrun the ranks rendezvous through the agent's store on MASTER_PORT; the children form
    # their own group, so child rank 0 must host a store of its own on the new port
    env["TORCHELASTIC_USE_AGENT_STORE"] = "False"
    cmd = [sys.executable, os.path.abspath(__file__), "--sg-child", "--gpus", str(world)]
    res = None
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.sg_timeout)
        if rank == 0:
            lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
            res = json.loads(lines[-1]) if p.returncode == 0 and lines else {"error": f"child rc={p.returncode}: {p.stderr[-200:]}"}
    except subprocess.TimeoutExpired:
        res = {"error": f"timeout after {args.sg_timeout:.0f} s (child killed)"}
    except Exception as exc:                                # noqa: BLE001
        res = {"error": repr(exc)[:200]}
    return res


def main():
    args = parse()
    if args.sg_child:
        return sg_child(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device; the HIP path has no CPU fallback")
    # IMGXF_BENCH_BACKEND=gloo + IMGXF_BENCH_SHARED_GPU=1: rehearsal of the N>1 control flow on a
    # one-GPU box (all ranks on device 0, CPU collectives); the driver's runs use nccl (= RCCL).
    # IMGXF_BENCH_FORCE_DIST=1: take the torch.distributed branch at world size 1 too (-m gpu test)
    backend = os.environ.get("IMGXF_BENCH_BACKEND", "nccl")
    if os.environ.get("IMGXF_BENCH_SHARED_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("IMGXF_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from imagetransformations_amd import ops   # loads libimgxf.so (raises if missing)

    F = args.frames
    precise = not args.fast_bilinear
    if args.only_1080p:                                      # (tools/profile_round.sh: FETCH_SIZE / WRITE_SIZE passes at 1080p)
        el, g2, r2, keep = timed_step(HHD, WHD, 4 * F, args, rank, world, dist, dev, backend, precise)
        if rank == 0:
            print(json.dumps({"only": "1920x1080", "frames_per_gpu": 4 * F, "ms_per_step": round(el / args.steps * 1e3, 4)}))
        if dist:
            dist.barrier(); dist.destroy_process_group()
        return
    sampler = SclkSampler(local)
    sampler.start()
    elapsed, g_ms, r_ms, keep = timed_step(H4K, W4K, F, args, rank, world, dist, dev, backend, precise)
    sclk = sampler.stop()
    frames, gaussian, rotate = keep
    sclk_kernel = in_kernel_sclk(dev, lambda: (gaussian(), rotate()))        # outside the timed region
    px_per_step = F * H4K * W4K
    value = world * px_per_step * args.steps / elapsed / 1e6

    gauss = kernel_entry("sepconv_march (5x5 Gaussian, 4K RGB)", GAUSS_BYTES_PER_PX, px_per_step, g_ms, pmc_traffic("sepconv", F))
    affine = kernel_entry("affine_bilinear_wq (rotate 30deg + 1.5x bilinear, 4K RGB)", AFFINE_BYTES_PER_PX, px_per_step, r_ms,
                          pmc_traffic("affine_bilinear", F))
    result = {
        "metric": METRIC,
        "value": round(value, 1), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/f32",
        "data": "synthetic",
        "config": {"workload": "5x5 separable Gaussian (sigma=5/6, REFLECT_101) + rotate 30deg/1.5x "
                               "bilinear (Pillow semantics) on 3840x2160 RGB uint8 frames",
                   "frames_per_gpu": F, "global_frames": F * world, "height": H4K, "width": W4K,
                   "bilinear": "fp64 bit-exact" if precise else "fp32",
                   "parallelism": f"frames sharded over {world} GPU(s), no data-path collective"},
        # headline kernel (north star: 5x5 Gaussian on 4K); every timed kernel is in roofline_kernels
        "roofline": gauss,
        "roofline_kernels": {"gaussian5x5_4k": gauss, "rotate30_zoom1.5_bilinear_4k": affine},
        "sclk_mhz": sclk, "sclk_in_kernel": sclk_kernel,
    }
    resolutions = {"3840x2160": {"value": round(value, 1), "unit": "Mpix/s", "ms_per_step": result["ms_per_step"],
                                 "frames_per_gpu": F, "gaussian_frac": gauss["frac"], "affine_frac": affine["frac"]}}
    del keep, frames, gaussian, rotate
    torch.cuda.empty_cache()

    if not args.no_1080p:
        FH = 4 * F                                           # same bytes per GPU as the 4K batch
        el, g2, r2, keep = timed_step(HHD, WHD, FH, args, rank, world, dist, dev, backend, precise)
        pxh = FH * HHD * WHD
        gh = kernel_entry("sepconv_march (5x5 Gaussian, 1080p RGB)", GAUSS_BYTES_PER_PX, pxh, g2, pmc_traffic("sepconv", F, "kernels_1080p"))
        ah = kernel_entry("affine_bilinear_wq (rotate 30deg + 1.5x bilinear, 1080p RGB)", AFFINE_BYTES_PER_PX, pxh, r2,
                          pmc_traffic("affine_bilinear", F, "kernels_1080p"))
        result["roofline_kernels"]["gaussian5x5_1080p"] = gh
        result["roofline_kernels"]["rotate30_zoom1.5_bilinear_1080p"] = ah
        resolutions["1920x1080"] = {"value": round(world * pxh * args.steps / el / 1e6, 1), "unit": "Mpix/s",
                                    "ms_per_step": round(el / args.steps * 1e3, 4), "frames_per_gpu": FH,
                                    "gaussian_frac": gh["frac"], "affine_frac": ah["frac"]}
        del keep
        torch.cuda.empty_cache()
    result["resolutions"] = resolutions

    if rank == 0 and not args.no_extras:
        result["ops"] = ops_extras(F, dev, sclk_kernel)

    # the sharded result is verified against the single-GPU one through a checksum (outside the timed region)
    result["checksum_ok"], result["checksum"] = verify_checksum(rank, world, dist, dev, precise)

    if world > 1 and not args.no_scatter_gather:
        # separate line: root -> ranks scatter and ranks -> root gather of 8 frames per
        # rank over RCCL point-to-point, in fresh children so that a stuck transfer is killed after a
        # timeout instead of hanging the headline
        sg = run_sg_children(args, rank, world, dist, dev, backend)
        if rank == 0:
            result["scatter_gather"] = sg

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"], result["cpu_baseline_all_cores"] = cpu_baselines()
    elif rank == 0:
        result["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(result))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
