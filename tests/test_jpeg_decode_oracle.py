"""CPU: the JPEG reader restatement (oracle/jpeg_decode_oracle.py) against Pillow / libjpeg-turbo, the codec behind
the reference's `Image.open(path).convert("RGB")` (/root/reference/transformation.py:83): bit-identical pixels on
files the reference itself wrote (tests/golden/reference_outputs/*.JPEG) and on seeded images of every size class,
chroma sampling, quality and table kind."""
import glob
import io
import os

import numpy as np
import pytest
from PIL import Image

from conftest import synth
from oracle import jpeg_decode_oracle as JD

REF = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "reference_outputs", "*.JPEG")))


def pillow_rgb(data: bytes) -> np.ndarray:
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


def photo_like(seed, h, w):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 128 + 70 * np.sin(xx / 9.0 + seed) + 50 * np.cos(yy / 7.0)
    img = base[..., None] + rng.normal(0, 12, (h, w, 3)) + np.array([10, -20, 30])
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("path", REF[::5], ids=lambda p: os.path.basename(p)[:40])
def test_reference_written_files(path):
    data = open(path, "rb").read()
    assert np.array_equal(JD.decode(data), pillow_rgb(data))


@pytest.mark.parametrize("h,w", [(1, 1), (7, 5), (8, 8), (16, 16), (17, 33), (31, 15), (48, 64), (100, 75),
                                 (9, 2), (40, 3), (174, 4), (21, 6), (2, 40)])    # widths <= 4: chroma narrower than 3 samples is replicated
@pytest.mark.parametrize("subsampling", [0, 1, 2])                      # 4:4:4, 4:2:2, 4:2:0
def test_sizes_and_samplings(h, w, subsampling):
    for seed, quality in ((1, 75), (2, 30), (3, 95)):
        for img in (synth(seed * 7 + h, h, w), photo_like(seed, h, w)):
            buf = io.BytesIO()
            Image.fromarray(img).save(buf, "JPEG", quality=quality, subsampling=subsampling)
            data = buf.getvalue()
            assert np.array_equal(JD.decode(data), pillow_rgb(data)), (h, w, subsampling, quality)


def test_grayscale_optimised_tables_and_restart_intervals():
    img = photo_like(5, 61, 83)
    for kwargs in (dict(optimize=True), dict(quality=10), dict(quality=100, subsampling=0), dict(restart_marker_blocks=3),
                   dict(restart_marker_rows=1, subsampling=2)):
        buf = io.BytesIO()
        try:
            Image.fromarray(img).save(buf, "JPEG", **kwargs)
        except TypeError:
            continue
        data = buf.getvalue()
        assert np.array_equal(JD.decode(data), pillow_rgb(data)), kwargs
    buf = io.BytesIO()
    Image.fromarray(img).convert("L").save(buf, "JPEG", quality=80)
    data = buf.getvalue()
    assert np.array_equal(JD.decode(data), pillow_rgb(data))


def test_unsupported_files_are_refused():
    buf = io.BytesIO()
    Image.fromarray(photo_like(1, 32, 32)).save(buf, "JPEG", progressive=True)
    with pytest.raises(JD.Unsupported):
        JD.decode(buf.getvalue())
    buf = io.BytesIO()
    Image.fromarray(photo_like(1, 32, 32)).convert("CMYK").save(buf, "JPEG")
    with pytest.raises(JD.Unsupported):
        JD.decode(buf.getvalue())
    with pytest.raises(JD.Unsupported):
        JD.decode(b"not a jpeg")
