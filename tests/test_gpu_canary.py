"""GPU: out-of-bounds canaries around every destination / workspace of the fast kernels
(SURVEY §5 "race detection / sanitizers": no GPU sanitizer exists on this pool).

Each case places the destination inside a larger allocation pre-filled with 0xA5: a lead-in, the
frames with padded rows (row_stride > row bytes) and padded frames (frame_stride > h*row_stride),
and a tail.  The op runs through the C-ABI on that strided view; afterwards every byte that is not
payload must still be 0xA5 and the payload must equal the same op's output on a dense destination.
Geometries target the kernels' edge handling: 1 KiB strips with 1040 / 1056 / 2048 / 2064-byte rows
(partial last strip, last lane, clamped re-reads), super-row frame seams, 1-row chunks, tile edges
of the affine kernels, odd sizes for the general paths."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import synth
from oracle import imgxf_oracle as O

pytestmark = pytest.mark.gpu
PAT = 0xA5


class Guarded:
    """A [n,h,w,c] uint8 (or `elem`-byte) view inside a 0xA5-filled allocation."""

    def __init__(self, device, n, h, w, c, row_pad=32, frame_pad=64, lead=256, elem=1):
        from imagetransformations_amd import _ffi as F
        self.n, self.h, self.w, self.c, self.elem = n, h, w, c, elem
        self.rb = w * c * elem
        self.rs = self.rb + row_pad
        self.fs = self.rs * h + frame_pad
        self.lead = lead
        self.total = 2 * lead + n * self.fs
        self.buf = torch.full((self.total,), PAT, dtype=torch.uint8, device=device)
        self.view = F.View(self.buf.data_ptr() + lead, n, h, w, c, self.rs, self.fs)

    def payload_mask(self):
        m = np.zeros(self.total, bool)
        for f in range(self.n):
            for y in range(self.h):
                a = self.lead + f * self.fs + y * self.rs
                m[a:a + self.rb] = True
        return m

    def check(self, want=None, what="", ties=False):
        """ties=True: the fp32 Gaussian kernels (marching / tiled) sum in different orders, so a value
        at a rounding boundary may land on either side: <= 1 LSB on < 1e-3 of the samples."""
        host = self.buf.cpu().numpy()
        m = self.payload_mask()
        bad = np.nonzero((host != PAT) & ~m)[0]
        assert bad.size == 0, f"{what}: {bad.size} guard bytes overwritten, first at offset {int(bad[0]) - self.lead} (rs={self.rs}, fs={self.fs})"
        if want is not None:
            got = host[m].reshape(np.asarray(want).shape) if self.elem == 1 else host[m].view(np.float32).reshape(np.asarray(want).shape)
            if ties:
                d = np.abs(got.astype(np.int32) - np.asarray(want).astype(np.int32))
                assert d.max() <= 1 and (d != 0).mean() < 1e-3, f"{what}: payload differs from the dense run beyond rounding ties"
            else:
                assert np.array_equal(got, np.asarray(want)), f"{what}: payload differs from the dense run"


def _dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _batch(seed, n, h, w, c=3):
    a = np.stack([synth(seed + i, h, w, c) for i in range(n)])
    return a[..., None] if c == 1 else a


GAUSS_GEOMS = [  # (n, h, w, c): row bytes 1056 / 2064 / 1040 / 2048 / 3840 / 183 (general path) / 1440
    (5, 37, 352, 3), (3, 21, 688, 3), (9, 18, 1040, 1), (4, 19, 512, 4), (8, 9, 1280, 3), (3, 37, 61, 3), (7, 5, 480, 3),
    (2, 1, 352, 3), (1, 3, 1360, 3)]


@pytest.mark.parametrize("geom", GAUSS_GEOMS)
@pytest.mark.parametrize("radius", [5 / 6, 1.5, 2.0, 5.0])
def test_gaussian_writes_only_its_destination(device, geom, radius):
    from imagetransformations_amd import _ffi as F, ops
    n, h, w, c = geom
    a = _batch(900, n, h, w, c)
    t = _dev(a, device)
    k = O.blur_ksize(radius)
    st = torch.cuda.current_stream().cuda_stream
    dense = ops.gaussian_blur(t, k, radius).cpu().numpy()
    for row_pad, frame_pad in ((32, 64), (16, 0), (0, 48), (5, 7)):          # the last one: unaligned strides -> general kernel
        g = Guarded(device, n, h, w, c, row_pad, frame_pad)
        F.call("imgxf_gaussian_u8", F.vp(F.view_of(t)), F.vp(g.view), k, float(radius), None, st)
        g.check(dense, f"gaussian k={k} pads=({row_pad},{frame_pad})", ties=True)
    # fp32 side output (4 bytes per sample) and the fixed-point instances
    g, gf = Guarded(device, n, h, w, c), Guarded(device, n, h, w, c, row_pad=64, frame_pad=128, elem=4)
    F.call("imgxf_gaussian_u8", F.vp(F.view_of(t)), F.vp(g.view), k, float(radius), F.vp(gf.view), st)
    g.check(dense, "gaussian + f32", ties=True); gf.check(None, "gaussian f32 side output")
    g = Guarded(device, n, h, w, c)
    F.call("imgxf_gaussian_cv_fixed_u8", F.vp(F.view_of(t)), F.vp(g.view), k, float(radius), st)
    g.check(ops.gaussian_blur(t, k, radius, fixed_point=True).cpu().numpy(), "gaussian fixed-point")


@pytest.mark.parametrize("geom", [(3, 270, 480), (2, 334, 500), (5, 96, 160), (2, 200, 352), (1, 270, 480)])
def test_affine_writes_only_its_destination(device, geom):
    from imagetransformations_amd import _ffi as F, ops
    n, h, w = geom
    a = _batch(910, n, h, w)
    t = _dev(a, device)
    st = torch.cuda.current_stream().cuda_stream
    fill = F.u8_array([3, 2, 1])
    for m, filt in ((O.rotate_zoom_matrix(w, h, 30.0, 1.5), F.FILTER_BILINEAR), (O.rotate_zoom_matrix(w, h, -70.0, 0.9), F.FILTER_BILINEAR),
                    (O.rotate_plan(w, h, 30.0)[1], F.FILTER_NEAREST), (O.shear_geometry(w, h, 0.3)[1], F.FILTER_BICUBIC)):
        ow = O.shear_geometry(w, h, 0.3)[0] if filt == F.FILTER_BICUBIC else w
        for precise in (1, 0):
            dense = torch.empty((n, h, ow, 3), dtype=torch.uint8, device=device)
            F.call("imgxf_affine_u8", F.vp(F.view_of(t)), F.vp(F.view_of(dense)), F.f64_array(m), filt, fill, precise, None, st)
            for row_pad, frame_pad in ((32, 64), (16, 0), (7, 5)):
                g = Guarded(device, n, h, ow, 3, row_pad, frame_pad)
                F.call("imgxf_affine_u8", F.vp(F.view_of(t)), F.vp(g.view), F.f64_array(m), filt, fill, precise, None, st)
                g.check(dense.cpu().numpy(), f"affine filter={filt} precise={precise} pads=({row_pad},{frame_pad})")


def test_sobel_mask_and_pointwise_write_only_their_destination(device):
    from imagetransformations_amd import _ffi as F, ops
    st = torch.cuda.current_stream().cuda_stream
    for n, h, w in ((3, 37, 352), (2, 9, 1040), (4, 30, 61), (1, 1, 1360)):
        a = _batch(920, n, h, w)
        t = _dev(a, device)
        for variant in (0, 1, 2):
            g = Guarded(device, n, h, w, 1, row_pad=16, frame_pad=32)
            F.call("imgxf_rgb_sobel_u8", F.vp(F.view_of(t)), F.vp(g.view), variant, st)
            g.check(ops.rgb_sobel(t, variant).cpu().numpy(), f"rgb_sobel variant {variant}")
        gray = ops.rgb2l(t)
        g = Guarded(device, n, h, w, 1, row_pad=16, frame_pad=0)
        F.call("imgxf_rgb2l_u8", F.vp(F.view_of(t)), F.vp(g.view), st)
        g.check(gray.cpu().numpy(), "rgb2l")
        g = Guarded(device, n, h, w, 1, row_pad=48, frame_pad=16)
        F.call("imgxf_sobel_u8", F.vp(F.view_of(gray)), F.vp(g.view), 0, st)
        g.check(ops.sobel(gray).cpu().numpy(), "sobel")
        mask = ops.percentile_mask(ops.sobel(gray), 70)
        g = Guarded(device, n, h, w, 1, row_pad=16, frame_pad=16)
        F.call("imgxf_dilate_cross_u8", F.vp(F.view_of(mask)), F.vp(g.view), 3, st)
        g.check(ops.dilate_cross(mask, 3).cpu().numpy(), "dilate")
        g = Guarded(device, n, h, w, 3)
        F.call("imgxf_scale_abs_u8", F.vp(F.view_of(t)), F.vp(g.view), 0.7, 0.0, st)
        g.check(ops.scale_abs(t, 0.7).cpu().numpy(), "scale_abs")
        g = Guarded(device, n, h, w, 3, row_pad=16)
        F.call("imgxf_blend_u8", None, F.u8_array([0, 0, 0]), F.vp(F.view_of(t)), None, F.vp(g.view), 1.05, st)
        g.check(ops.brightness(t, 1.05).cpu().numpy(), "brightness")
        g = Guarded(device, n, h, w, 3, row_pad=16)
        F.call("imgxf_composite_const_u8", F.vp(F.view_of(t)), F.u8_array([255, 0, 0]), F.vp(F.view_of(mask)), F.vp(g.view), st)
        g.check(ops.composite_const(t, (255, 0, 0), mask).cpu().numpy(), "composite")


@pytest.mark.parametrize("scale", [1.1, 0.9, 1.3])
def test_fused_resample_writes_only_destination(device, scale):
    """The matrix-core kernel: padded destination rows, no workspace at all (NULL is accepted)."""
    from imagetransformations_amd import _ffi as F, ops
    st = torch.cuda.current_stream().cuda_stream
    n, h, w = 3, 135, 240
    t = _dev(_batch(931, n, h, w), device)
    nw, nh = int(w * scale), int(h * scale)
    dense = ops.resize_lanczos(t, (nw, nh)).cpu().numpy()
    plan = ctypes.c_void_p()
    F.call("imgxf_resample_plan_create", ctypes.byref(plan), h, w, nh, nw, 3, 0, 1)
    try:
        k, nbytes = ctypes.c_int(), ctypes.c_size_t(1)
        F.call("imgxf_resample_plan_kernel", plan, ctypes.byref(k))
        assert k.value > 0
        g = Guarded(device, n, nh, nw, 3, row_pad=16, frame_pad=32)
        F.call("imgxf_resample_workspace_bytes_for", plan, F.vp(F.view_of(t)), F.vp(g.view), ctypes.byref(nbytes))
        assert nbytes.value == 0
        F.call("imgxf_resample_ws_u8", plan, F.vp(F.view_of(t)), F.vp(g.view), None, 0, st)
        g.check(dense, "fused lanczos")
    finally:
        F.call("imgxf_lanczos_plan_destroy", plan)


@pytest.mark.parametrize("scale", [1.1, 0.9, 1.3])
def test_resample_writes_only_destination_and_workspace(device, scale, monkeypatch):
    from imagetransformations_amd import _ffi as F, ops
    monkeypatch.setenv("IMGXF_RESAMPLE_NO_MFMA", "1")       # the two-pass kernels and their intermediate
    st = torch.cuda.current_stream().cuda_stream
    n, h, w = 3, 135, 240
    t = _dev(_batch(930, n, h, w), device)
    nw, nh = int(w * scale), int(h * scale)
    dense = ops.resize_lanczos(t, (nw, nh)).cpu().numpy()
    plan = ctypes.c_void_p()
    F.call("imgxf_resample_plan_create", ctypes.byref(plan), h, w, nh, nw, 3, 0, 1)
    try:
        nbytes = ctypes.c_size_t()
        F.call("imgxf_resample_workspace_bytes", plan, n, ctypes.byref(nbytes))
        ws = torch.full((nbytes.value + 512,), PAT, dtype=torch.uint8, device=device)
        g = Guarded(device, n, nh, nw, 3, row_pad=16, frame_pad=32)
        F.call("imgxf_resample_ws_u8", plan, F.vp(F.view_of(t)), F.vp(g.view), ws.data_ptr() + 256, nbytes.value, st)
        g.check(dense, "lanczos")
        hw = ws.cpu().numpy()
        assert (hw[:256] == PAT).all() and (hw[256 + nbytes.value:] == PAT).all(), "workspace overrun"
        with pytest.raises(Exception):                      # too small a workspace is refused, nothing runs
            F.call("imgxf_resample_ws_u8", plan, F.vp(F.view_of(t)), F.vp(g.view), ws.data_ptr(), max(nbytes.value - 1, 0), st)
    finally:
        F.call("imgxf_lanczos_plan_destroy", plan)


def test_resample_plan_is_shared_across_streams(device):
    """one cached plan, two torch streams, same geometry: the per-call workspace keeps the
    results apart (the shared-intermediate plan raced here)."""
    from imagetransformations_amd import ops
    a = _dev(_batch(940, 6, 270, 480), device)
    b = _dev(_batch(950, 6, 270, 480), device)
    wa, wb = ops.resize_lanczos(a, (528, 297)), ops.resize_lanczos(b, (528, 297))
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(5):
        with torch.cuda.stream(s1):
            ra = ops.resize_lanczos(a, (528, 297))
        with torch.cuda.stream(s2):
            rb = ops.resize_lanczos(b, (528, 297))
        torch.cuda.synchronize()
        assert torch.equal(ra, wa) and torch.equal(rb, wb)
    # a larger batch through the same cached plan (the old cache destroyed and re-created it)
    big = torch.cat([a, b])
    assert torch.equal(ops.resize_lanczos(big, (528, 297)), torch.cat([wa, wb]))
