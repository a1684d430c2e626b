"""The oracle (CPU) and the HIP path (GPU) against the reference pipeline's OWN outputs.

`tests/golden/reference_outputs/` holds JPEG files the reference wrote
(/root/reference/transformation.py:161-162).  The originals are not available, but four images
have an identity-parameter output (contrast 1.0, noise 0.0, rotation 0.0) = the original after
one JPEG round trip.  Applying transformation T with the file name's parameter to that proxy
must reproduce the reference's output of T up to JPEG noise: exact output size, PSNR well above
what any wrong convention (rotation sign, shear direction, translation sign, fill colour,
blur kernel-size rule, contrast / brightness law) gives.  This is the only anchor that exists
for the OpenCV-backed transforms (blur, contrast), whose library is not installed here.
"""
import io
import os
import re

import numpy as np
import pytest

from oracle import imgxf_oracle as O

Image = pytest.importorskip("PIL.Image")
DIR = os.path.join(os.path.dirname(__file__), "golden", "reference_outputs")
IDENTITY = {"ILSVRC2012_val_00005548": "contrast_1.0", "ILSVRC2012_val_00017407": "contrast_1.0",
            "ILSVRC2012_val_00035433": "gaussian_noise_0.0", "ILSVRC2012_val_00048138": "rotation_0.0"}
TYPES = ("lighten_darken", "gaussian_noise", "translation", "rotation", "contrast", "scale", "shear", "blur")


def load(name):
    return np.asarray(Image.open(os.path.join(DIR, name)).convert("RGB"))


def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


def jpeg_roundtrip(a):
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, format="JPEG")          # Image.save defaults, as the reference
    return np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))


def cases():
    out = []
    for f in sorted(os.listdir(DIR)):
        m = re.match(r"(ILSVRC2012_val_\d+)_(.+)_corrupted\.JPEG", f)
        if not m:
            continue
        name, rest = m.groups()
        for t in TYPES:
            if rest.startswith(t + "_"):
                out.append((name, t, rest[len(t) + 1:], f))
    return out


def apply(impl, t, value, x):
    """impl: oracle module (arrays) or the GPU facade (PIL images)."""
    if t == "translation":
        tx, ty = value.split("_")
        return impl("translation", x, float(tx), float(ty))
    return impl(t, x, float(value))


def oracle_impl(t, x, *args):
    fn = {"translation": O.apply_translation, "lighten_darken": O.apply_brightness, "rotation": O.apply_rotation,
          "contrast": O.apply_contrast, "scale": O.apply_scale, "shear": O.apply_shear, "blur": O.apply_blur}[t]
    return fn(x, *args)


# what a wrong convention would look like, per transform (must score clearly worse)
WRONG = {
    "rotation": lambda impl, x, v: impl("rotation", x, -v),
    "shear": lambda impl, x, v: impl("shear", x[:, ::-1].copy(), v)[:, ::-1],
    "blur": lambda impl, x, v: impl("blur", x, v / 2.5),
    "contrast": lambda impl, x, v: impl("lighten_darken", x, v - 1.0 + 0.1),
    "lighten_darken": lambda impl, x, v: impl("lighten_darken", x, -v),
    "scale": lambda impl, x, v: impl("scale", x, 2.0 - v),
}


def check(impl, name, t, value, fname):
    proxy = load(f"{name}_{IDENTITY[name]}_corrupted.JPEG")
    ref = load(fname)
    if t == "gaussian_noise":
        sigma = float(value)
        assert ref.shape == proxy.shape
        measured = (ref.astype(np.float64) - proxy).std() / 255.0
        assert (sigma == 0.0 and measured == 0.0) or 0.3 * sigma <= measured <= 1.1 * sigma
        return
    out = np.asarray(apply(impl, t, value, proxy))
    assert out.shape == ref.shape, (t, value, out.shape, ref.shape)         # incl. shear's w + ceil(sh*h)
    good = psnr(jpeg_roundtrip(out), ref)
    floor_db = 40.0 if t == "blur" else 29.0
    assert good >= floor_db, (t, value, good)
    if t == "translation":
        tx, ty = (float(v) for v in value.split("_"))
        if tx or ty:
            bad = psnr(jpeg_roundtrip(np.asarray(impl("translation", proxy, -tx, -ty))), ref)
            assert good >= bad + 8.0, (t, value, good, bad)
    elif t in WRONG and float(value) not in (0.0, 1.0):
        bad_out = np.asarray(WRONG[t](impl, proxy, float(value)))
        if bad_out.shape == ref.shape:
            bad = psnr(jpeg_roundtrip(bad_out), ref)
            assert good >= bad + (3.0 if t == "lighten_darken" else 5.0), (t, value, good, bad)
    if t in ("rotation", "scale") and float(value) not in (0.0, 1.0):
        # fill convention: black corners where the source does not reach
        if t == "rotation" or float(value) < 1.0:
            assert ref[0, 0].max() <= 12 and out[0, 0].max() == 0
    if t == "shear":
        # white fill in the triangles the sheared source does not reach (top-left, bottom-right)
        assert ref[2:8, 2:8].min() >= 240 and out[2:8, 2:8].min() == 255
        assert ref[-8:-2, -8:-2].min() >= 240 and out[-8:-2, -8:-2].min() == 255


@pytest.mark.parametrize("name,t,value,fname", cases())
def test_oracle_reproduces_the_reference_outputs(name, t, value, fname):
    check(oracle_impl, name, t, value, fname)


@pytest.mark.gpu
@pytest.mark.parametrize("name,t,value,fname", cases())
def test_hip_path_reproduces_the_reference_outputs(device, name, t, value, fname):
    from imagetransformations_amd import transformation as T
    fns = {"translation": T.apply_translation, "lighten_darken": T.apply_brightness, "rotation": T.apply_rotation,
           "contrast": T.apply_contrast, "scale": T.apply_scale, "shear": T.apply_shear, "blur": T.apply_blur}

    def impl(t_, x, *args):
        return np.asarray(fns[t_](Image.fromarray(np.ascontiguousarray(x)), *args))
    check(impl, name, t, value, fname)


@pytest.mark.gpu
def test_hip_fixed_point_blur_is_at_least_as_close_to_the_reference_outputs(device):
    """The opt-in fixed-point Gaussian (transformation.BLUR_FIXED_POINT) against the reference's
    blur outputs: never worse than the float definition by more than JPEG noise, better on average."""
    from imagetransformations_amd import transformation as T
    gains = []
    for name, t, value, fname in cases():
        if t != "blur" or float(value) == 0.0:
            continue
        ref, proxy = load(fname), load(f"{name}_{IDENTITY[name]}_corrupted.JPEG")
        img = Image.fromarray(proxy)
        flt = psnr(jpeg_roundtrip(np.asarray(T.apply_blur(img, float(value)))), ref)
        try:
            T.BLUR_FIXED_POINT = True
            fix = psnr(jpeg_roundtrip(np.asarray(T.apply_blur(img, float(value)))), ref)
        finally:
            T.BLUR_FIXED_POINT = False
        assert fix >= flt - 0.3, (fname, fix, flt)
        gains.append(fix - flt)
    if gains:
        assert np.mean(gains) >= 0.0, gains
