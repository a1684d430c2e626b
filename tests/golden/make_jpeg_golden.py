"""Writes tests/golden/jpeg_q75.json: sha256 of Pillow's own `Image.save(fp, "JPEG")` output (the library call behind
the reference's `transformed.save(path)`, transformation.py:161-162) for seeded synthetic RGB images.
usage: python tests/golden/make_jpeg_golden.py"""
import hashlib, io, json, os
import numpy as np
import PIL
from PIL import Image, features

CASES = [(16, 16), (8, 8), (1, 1), (17, 33), (24, 40), (9, 23), (31, 17), (48, 64), (15, 16), (16, 15), (33, 47), (375, 500)]


def image(h, w, kind, seed):
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(xx * 5 + yy * 3) % 256, (xx * 2 + yy * 7) % 256, ((xx + yy) * 2) % 256], -1)
    return ((base + rng.integers(0, 6, (h, w, 3))) % 256).astype(np.uint8)


if __name__ == "__main__":
    rows = []
    for i, (h, w) in enumerate(CASES):
        for kind in ("noise", "smooth"):
            for q in (None, 90):
                buf = io.BytesIO()
                Image.fromarray(image(h, w, kind, 100 + i)).save(buf, "JPEG", **({} if q is None else {"quality": q}))
                rows.append({"h": h, "w": w, "kind": kind, "seed": 100 + i, "quality": q, "bytes": len(buf.getvalue()),
                             "sha256": hashlib.sha256(buf.getvalue()).hexdigest()})
    out = {"pillow": PIL.__version__, "libjpeg": features.version("jpg"), "turbo": features.check_feature("libjpeg_turbo"), "cases": rows}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "jpeg_q75.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(len(rows), "cases")
