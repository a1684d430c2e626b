"""Same-process A/B of two builds of libimgxf.so on the headline kernels (development aid).

    python tools/ab_lib.py <libA.so> <libB.so> [affine|affine32|gaussian|nearest|all] [frames]

Both libraries are loaded with ctypes next to each other; the same C-ABI call on the same resident
frames is timed with HIP events, alternating A / B for ROUNDS rounds of ITERS launches, and the medians are
printed (box-to-box and process-to-process differences cancel).  Outputs are compared byte for byte."""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from imagetransformations_amd import _ffi, ops  # noqa: E402

ROUNDS, ITERS = int(os.environ.get("ROUNDS", "7")), int(os.environ.get("ITERS", "8"))


def load(path, env=None):
    """env: "NAME=value,NAME2=value" knobs that only THIS library instance sees (every copy of libimgxf.so caches its own
    knob table: the variables are set, the library re-reads them, they are removed again)."""
    if env:
        import shutil, tempfile
        tmp = tempfile.NamedTemporaryFile(suffix=".so", delete=False).name       # a second dlopen of one path is the same instance
        shutil.copy(path, tmp)
        path = tmp
    lib = C.CDLL(path)
    if env:
        pairs = [kv.split("=", 1) for kv in env.split(",")]
        for k, v in pairs: os.environ[k] = v
        lib.imgxf_reload_knobs()
        for k, _ in pairs: os.environ.pop(k)
    for name, argtypes in _ffi.SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = C.c_int, argtypes
    return lib


def main():
    pa, pb = sys.argv[1], sys.argv[2]
    what = sys.argv[3] if len(sys.argv) > 3 else "all"
    F = int(sys.argv[4]) if len(sys.argv) > 4 else 128
    libs = {"A": load(pa, os.environ.get("A_ENV")), "B": load(pb, os.environ.get("B_ENV"))}
    H, W = 2160, 3840
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    src = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device=dev, generator=gen)
    outs = {k: torch.empty_like(src) for k in libs}
    st = torch.cuda.current_stream().cuda_stream
    vs = _ffi.view_of(src)
    m = _ffi.f64_array(ops.rotate_zoom_matrix(W, H, 30.0, 1.5))
    mn = _ffi.f64_array(ops.rotate_matrix(W, H, 22.5))
    fill = _ffi.u8_array([0, 0, 0])
    cases = {
        "affine": (lambda lib, vo: lib.imgxf_affine_u8(_ffi.vp(vs), _ffi.vp(vo), m, 1, fill, 1, None, st), 4.306),
        "affine32": (lambda lib, vo: lib.imgxf_affine_u8(_ffi.vp(vs), _ffi.vp(vo), m, 1, fill, 0, None, st), 4.306),
        "gaussian": (lambda lib, vo: lib.imgxf_gaussian_u8(_ffi.vp(vs), _ffi.vp(vo), 5, 5.0 / 6.0, None, st), 6.0),
        "nearest": (lambda lib, vo: lib.imgxf_affine_u8(_ffi.vp(vs), _ffi.vp(vo), mn, 0, fill, 1, None, st), 5.06),
        "gaussian31": (lambda lib, vo: lib.imgxf_gaussian_u8(_ffi.vp(vs), _ffi.vp(vo), 31, 5.0, None, st), 6.0),
    }
    if what == "step":
        # bench.py's step: Gaussian (src -> tmp) then rotate (tmp -> out), alternating libraries per round; per-kernel events
        tmp = torch.empty_like(src)
        vt = _ffi.view_of(tmp)
        res = {k: ([], []) for k in libs}
        for rnd in range(ROUNDS + 1):
            for k, lib in libs.items():
                vo = _ffi.view_of(outs[k])
                ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(ITERS)]
                for it in range(ITERS):
                    ev[it][0].record()
                    lib.imgxf_gaussian_u8(_ffi.vp(vs), _ffi.vp(vt), 5, 5.0 / 6.0, None, st)
                    ev[it][1].record()
                    lib.imgxf_affine_u8(_ffi.vp(vt), _ffi.vp(vo), m, 1, fill, 1, None, st)
                    ev[it][2].record()
                torch.cuda.synchronize()
                if rnd:
                    res[k][0].extend(e[0].elapsed_time(e[1]) for e in ev)
                    res[k][1].extend(e[1].elapsed_time(e[2]) for e in ev)
        px = F * H * W
        for k in libs:
            g, a = statistics.median(res[k][0]), statistics.median(res[k][1])
            print(f"step, library {k}: gaussian {g:7.4f} ms ({6.0 * px / g / 1e6 / 8000:.3f})  rotate {a:7.4f} ms ({4.306 * px / a / 1e6 / 8000:.3f})  "
                  f"step {g + a:7.4f} ms = {px / (g + a) / 1e3:9.1f} Mpix/s", flush=True)
        print("outputs equal:", bool(torch.equal(outs["A"], outs["B"])))
        return
    names = list(cases) if what == "all" else what.split(",")
    px = F * H * W
    for name in names:
        call, bpp = cases[name]
        res = {k: [] for k in libs}
        views = {k: _ffi.view_of(outs[k]) for k in libs}
        for k, lib in libs.items():
            rc = call(lib, views[k])
            assert rc == 0, (name, k, rc)
        torch.cuda.synchronize()
        equal = bool(torch.equal(outs["A"], outs["B"]))
        for _ in range(ROUNDS):
            for k, lib in libs.items():
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                call(lib, views[k]); s.record()
                for _ in range(ITERS):
                    call(lib, views[k])
                e.record(); torch.cuda.synchronize()
                res[k].append(s.elapsed_time(e) / ITERS)
        a, b = statistics.median(res["A"]), statistics.median(res["B"])
        print(f"{name:10s} F={F}  A {a:7.4f} ms ({bpp * px / a / 1e6 / 8000:.3f})   B {b:7.4f} ms ({bpp * px / b / 1e6 / 8000:.3f})"
              f"   B/A {b / a:.3f}   outputs equal: {equal}", flush=True)


if __name__ == "__main__":
    main()
