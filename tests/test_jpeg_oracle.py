"""The JPEG-writer restatement (oracle/jpeg_oracle.py) against the committed Pillow fixtures and, where Pillow is
importable, against Pillow itself (the library the reference's `transformed.save(path)` calls, transformation.py:161-162)."""
import hashlib, importlib.util, io, json, os
import numpy as np
import pytest
from oracle import jpeg_oracle as J

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_jpeg_golden", os.path.join(HERE, "golden", "make_jpeg_golden.py"))
G = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(G)
GOLDEN = json.load(open(os.path.join(HERE, "golden", "jpeg_q75.json")))


@pytest.mark.parametrize("case", GOLDEN["cases"], ids=lambda c: f"{c['h']}x{c['w']}-{c['kind']}-q{c['quality']}")
def test_oracle_reproduces_pillow_fixture(case):
    img = G.image(case["h"], case["w"], case["kind"], case["seed"])
    out = J.encode(img, 75 if case["quality"] is None else case["quality"])
    assert len(out) == case["bytes"]
    assert hashlib.sha256(out).hexdigest() == case["sha256"]


@pytest.mark.parametrize("quality", [1, 10, 50, 75, 95, 100])
@pytest.mark.parametrize("shape", [(17, 33), (40, 56), (2, 2)])
def test_oracle_equals_pillow(quality, shape):
    from PIL import Image
    rng = np.random.default_rng(quality * 1000 + shape[0])
    img = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    if shape[0] == 40:
        img = (img // 32 * 8 + 100).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, "JPEG", quality=quality)
    assert J.encode(img, quality) == buf.getvalue()


def test_header_layout():
    hdr = J.header(40, 24, J.quant_tables(75))
    assert len(hdr) == 623 and hdr[:4] == b"\xff\xd8\xff\xe0" and hdr[-14:-12] == b"\xff\xda"
