"""GPU: apply_perspective_warp (fall_2025/transformations_code:54-66) and its kernel through the
C-ABI against the oracle, which tests/test_oracle_vs_libs.py pins bit-for-bit on torch's CPU
grid_sample.  fp32 path: the kernel repeats the CPU build's operation order, so the comparison
is exact; the tolerance the contract would allow (1 LSB from the final truncation) is not used."""
import random

import numpy as np
import pytest
import torch

from conftest import synth
from oracle import imgxf_oracle as O

pytestmark = pytest.mark.gpu
Image = pytest.importorskip("PIL.Image")


def _coeffs(w, h, ds, seed):
    g = torch.Generator().manual_seed(seed)
    st, en = O.perspective_endpoints(w, h, ds, lambda lo, hi: int(torch.randint(lo, hi, size=(1,), generator=g).item()))
    return [float(v) for v in O.perspective_coeffs(st, en)]


@pytest.mark.parametrize("hw", [(32, 32), (37, 61), (334, 500), (200, 3), (5, 301), (1, 1), (129, 257)])
def test_kernel_matches_oracle(device, hw):
    from imagetransformations_amd import ops
    h, w = hw
    for ch in (3, 1, 4):
        a = synth(7 + ch, h, w, ch)
        for k, ds in enumerate((0.0, 0.05, 0.2, 0.6, 1.0)):
            c = _coeffs(w, h, ds, 10 * k + ch) if min(h, w) > 1 else [1.0 + ds, 0.1, -0.3, 0.0, 1.0, 0.2 * ds, 0.01, 0.0]
            got = ops.perspective(torch.from_numpy(a).to(device), c).cpu().numpy()
            assert np.array_equal(got, O.perspective_warp(a, c)), (hw, ch, ds)


def test_per_frame_coefficients_and_strided_batches(device):
    from imagetransformations_amd import ops
    h, w, n = 70, 90, 53                      # more frames than one launch carries coefficients for
    frames = np.stack([synth(200 + i, h, w) for i in range(n)])
    cs = [_coeffs(w, h, (0.1, 0.2, 0.4)[i % 3], i) for i in range(n)]
    t = torch.from_numpy(frames).to(device)
    got = ops.perspective(t, cs).cpu().numpy()
    for i in range(n):
        assert np.array_equal(got[i], O.perspective_warp(frames[i], cs[i])), i
    shared = ops.perspective(t, cs[1]).cpu().numpy()
    for i in (0, 17, 52):
        assert np.array_equal(shared[i], O.perspective_warp(frames[i], cs[1]))
    wide = torch.zeros((n, h, w + 11, 3), dtype=torch.uint8, device=device)
    wide[:, :, 4:4 + w] = t
    view = wide[::2, :, 4:4 + w]              # frame stride 2 images, row stride wider than a row
    got = ops.perspective(view, cs[2]).cpu().numpy()
    for j, i in enumerate(range(0, n, 2)):
        assert np.array_equal(got[j], O.perspective_warp(frames[i], cs[2])), i


def test_extreme_coefficients(device):
    """Maps the staged-box argument does not cover (denominator changing sign inside the image,
    strong minification, everything outside) take the global-memory path; same bytes."""
    from imagetransformations_amd import ops
    h, w = 96, 160
    a = synth(5, h, w)
    t = torch.from_numpy(a).to(device)
    cases = [
        [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, -0.0125, 0.0],          # denominator zero at x = 80
        [1.0, 0.2, 3.0, -0.1, 1.0, 1.0, 0.004, -0.02],
        [9.0, 0.0, -300.0, 0.0, 7.0, -200.0, 0.0, 0.0],        # 9x minification: boxes too large for LDS
        [1.0, 0.0, 5000.0, 0.0, 1.0, 0.0, 0.0, 0.0],           # entirely outside
        [0.0, 0.0, 10.0, 0.0, 0.0, 20.0, 0.0, 0.0],            # constant source position
        [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0],              # identity
        [-1.0, 0.0, 159.0, 0.0, -1.0, 95.0, 0.0, 0.0],         # point reflection
        [1e-3, 0.0, 0.5, 0.0, 1e-3, 0.5, 0.0, 0.0],
    ]
    for c in cases:
        got = ops.perspective(t, c).cpu().numpy()
        with np.errstate(all="ignore"):
            want = O.perspective_warp(a, c)
        assert np.array_equal(got, want), c


def test_full_hd_frame(device):
    from imagetransformations_amd import ops
    h, w = 1080, 1920
    a = synth(77, h, w)
    c = _coeffs(w, h, 0.2, 3)
    got = ops.perspective(torch.from_numpy(a).to(device), c).cpu().numpy()
    assert np.array_equal(got, O.perspective_warp(a, c))


def test_facade_draws_and_pixels(device):
    """apply_perspective_warp(img, v): same torch generator stream as RandomPerspective(v, p=1)."""
    from imagetransformations_amd import transformations_code as TC
    for seed, hw in enumerate([(32, 32), (61, 37), (334, 500)]):
        a = synth(seed, *hw)
        img = Image.fromarray(a)
        for ds in (0.0, 0.05, 0.1, 0.15000000000000002, 0.2):
            torch.manual_seed(seed)
            got = np.asarray(TC.apply_perspective_warp(img, ds))
            torch.manual_seed(seed)
            torch.rand(1)
            st, en = O.perspective_endpoints(hw[1], hw[0], ds, lambda lo, hi: int(torch.randint(lo, hi, size=(1,)).item()))
            c = [float(v) for v in O.perspective_coeffs(st, en)]
            want = O.perspective_warp(a, c)
            # the host solves with torch.linalg.lstsq (as torchvision does), the oracle with an 8x8
            # solve: the fp32 coefficients may differ in the last bit, which can move a byte by 1
            diff = np.abs(got.astype(int) - want.astype(int))
            assert diff.max() <= 1 and (diff != 0).mean() < 2e-3, (hw, ds, diff.max(), (diff != 0).mean())
    grey = Image.fromarray(synth(9, 40, 50)[..., 0].copy())
    torch.manual_seed(1)
    out = TC.apply_perspective_warp(grey, 0.2)
    assert out.mode == "L" and out.size == grey.size


def test_twelve_transformation_driver(device):
    """fall_2025/transformations_code:68-155: twelve outputs per image, the reference's names."""
    from imagetransformations_amd import transformations_code as TC
    imgs = [(Image.fromarray(synth(40 + i, 32, 32)), f"cifar10_test_{i}_label_{i % 10}") for i in range(3)]
    random.seed(4); np.random.seed(4); torch.manual_seed(4)
    out = TC.apply_all_transformations(imgs)
    assert len(out) == 12 * len(imgs)
    keys = list(TC.TRANSFORMATIONS_2D)
    assert all(o.mode == "RGB" and (o.size == (32, 32) or keys[j % 12] == 'shear') for j, o in enumerate(out))
    # replay the draws: the flip and every deterministic body can be checked per pixel
    random.seed(4)
    for i, (img, name) in enumerate(imgs):
        a = np.asarray(img)
        vals = {}
        for t, prm in TC.TRANSFORMATIONS_2D.items():
            if 'apply' in prm:
                continue
            steps = int((prm['max'] - prm['min']) / prm['step']) + 1
            grid = [prm['min'] + j * prm['step'] for j in range(steps)]
            vals[t] = (random.choice(grid), random.choice(grid)) if t == 'translation' else random.choice(grid)
        res = dict(zip(keys, out[12 * i:12 * i + 12]))
        assert np.array_equal(np.asarray(res['scale']), O.apply_scale(a, vals['scale']))
        assert np.array_equal(np.asarray(res['rotation']), O.apply_rotation(a, vals['rotation']))
        assert np.array_equal(np.asarray(res['shear']), O.apply_shear(a, vals['shear']))
        assert np.array_equal(np.asarray(res['zoom']), O.apply_scale(a, vals['zoom']))
        assert np.array_equal(np.asarray(res['vert_flip']), a[:, ::-1])
        assert np.array_equal(np.asarray(res['translation']), O.apply_translation(a, *vals['translation']))


def test_error_paths(device):
    from imagetransformations_amd import ops
    t = torch.zeros((2, 8, 8, 3), dtype=torch.uint8, device=device)
    ident = [1.0, 0, 0, 0, 1.0, 0, 0, 0]
    with pytest.raises(ValueError):
        ops.perspective(t, ident[:7])
    with pytest.raises(ValueError):
        ops.perspective(t, [ident])                       # one row for two frames
    with pytest.raises(Exception):
        ops.perspective(t, [float("nan")] + ident[1:])
    assert ops.perspective(t[:0], ident).shape == (0, 8, 8, 3)


def test_batched_twelve_transformation_driver_equals_per_image_driver(device):
    """Same `random` / `np.random` / `torch` draws, same names and order, same pixels as the
    per-image loop (fall_2025/transformations_code:68-155), for mixed image sizes."""
    from imagetransformations_amd import transformations_code as TC
    imgs = [(Image.fromarray(synth(70 + i, *hw)), f"cifar10_test_{i}_label_{i % 10}")
            for i, hw in enumerate([(32, 32), (64, 64), (32, 32), (50, 61), (64, 64), (32, 32), (32, 32)])]   # rand_crop needs h >= int(0.78 w)
    for seed in (0, 1):
        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
        a = TC.apply_all_transformations(imgs)
        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
        b = TC.apply_all_transformations_batched(imgs)
        assert len(a) == len(b) == 12 * len(imgs)
        for j, (x, y) in enumerate(zip(a, b)):
            assert x.size == y.size and x.mode == y.mode, j
            assert np.array_equal(np.asarray(x), np.asarray(y)), (seed, j, list(TC.TRANSFORMATIONS_2D)[j % 12])
    # both drivers leave the three generators in the same state
    random.seed(5); np.random.seed(5); torch.manual_seed(5)
    TC.apply_all_transformations(imgs)
    tail_a = (random.random(), float(np.random.rand()), float(torch.rand(1)))
    random.seed(5); np.random.seed(5); torch.manual_seed(5)
    TC.apply_all_transformations_batched(imgs)
    assert tail_a == (random.random(), float(np.random.rand()), float(torch.rand(1)))


def test_new_entry_points_are_stream_capturable(device):
    """perspective warp (per-frame coefficients travel as kernel arguments), the fixed-point
    Gaussian and ToTensor+Normalize captured into one HIP graph and replayed on new input."""
    from imagetransformations_amd import _ffi
    n, h, w = 6, 48, 64
    batch = np.stack([synth(300 + i, h, w) for i in range(n)])
    src = torch.from_numpy(batch).to(device)
    a, b = torch.empty_like(src), torch.empty_like(src)
    out = torch.empty((n, 3, h, w), dtype=torch.float32, device=device)
    coeffs = _ffi.f32_array([v for i in range(n) for v in _coeffs(w, h, 0.2, i)])
    mean, std = _ffi.f32_array([0.5, 0.4, 0.3]), _ffi.f32_array([0.2, 0.25, 0.3])

    def chain(stream):
        _ffi.call("imgxf_perspective_bilinear_u8", _ffi.vp(_ffi.view_of(src)), _ffi.vp(_ffi.view_of(a)), coeffs, 1, stream)
        _ffi.call("imgxf_gaussian_cv_fixed_u8", _ffi.vp(_ffi.view_of(a)), _ffi.vp(_ffi.view_of(b)), 5, 5.0 / 6.0, stream)
        _ffi.call("imgxf_to_tensor_f32", _ffi.vp(_ffi.view_of(b)), out.data_ptr(), mean, std, stream)

    chain(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = out.clone()
    out.zero_()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            chain(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert not torch.equal(out, want)
    g.replay(); torch.cuda.synchronize()
    assert torch.equal(out, want)
    src.copy_(torch.from_numpy(batch[::-1].copy()).to(device))
    g.replay(); torch.cuda.synchronize()
    got = out.clone()
    chain(torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    assert torch.equal(got, out) and not torch.equal(got, want)
