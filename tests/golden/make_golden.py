#!/usr/bin/env python3
"""Generate tests/golden/*.npz — golden vectors for the hot path.

Run in the build container only (needs Pillow + SciPy):  python tests/golden/make_golden.py

Every expected output is produced by calling the THIRD-PARTY library entry the reference
calls, with the reference's own argument list (file:line cited per case); the reference
module itself cannot be imported here (it imports cv2, which is not installed, at
/root/reference/transformation.py:7).  Inputs are synthetic: default_rng(seed).integers(0,256).
cv2-backed ops (GaussianBlur, convertScaleAbs, filter2D) have NO library here: their vectors
come from the float definition in oracle/imgxf_oracle.py and are marked `pinned=False`.
"""
import hashlib
import os
import sys

import numpy as np
from PIL import Image, ImageEnhance
from scipy import ndimage
from scipy.ndimage import binary_dilation

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import imgxf_oracle as O  # noqa: E402  (only for the unpinned cv2 definitions + grids)

SIZES = [(32, 32), (37, 61), (48, 64)]


def synth(seed, h, w, c=3):
    shape = (h, w) if c == 1 else (h, w, c)
    return np.random.default_rng(seed).integers(0, 256, shape, dtype=np.uint8)


def ref_scale(img, s):                       # transformation.py:173-196
    w, h = img.size
    nw, nh = int(w * s), int(h * s)
    sc = img.resize((nw, nh), Image.Resampling.LANCZOS)
    if s > 1.0:
        l, t = (nw - w) // 2, (nh - h) // 2
        sc = sc.crop((l, t, l + w, t + h))
    elif s < 1.0:
        r = Image.new('RGB', (w, h), (0, 0, 0))
        r.paste(sc, ((w - nw) // 2, (h - nh) // 2))
        sc = r
    return sc


def ref_translation(img, tx, ty):            # transformation.py:284-307
    w, h = img.size
    res = Image.new('RGB', (w, h), (0, 0, 0))
    px, py = int(tx), int(ty)
    cl, ct = max(0, -px), max(0, -py)
    cr, cb = min(w, w - px), min(h, h - py)
    if cl < cr and ct < cb:
        res.paste(img.crop((cl, ct, cr, cb)), (max(0, px), max(0, py)))
    return res


def ref_shear(img, sh):                      # transformation.py:212-226
    import math
    w, h = img.size
    shift = int(math.ceil(sh * h))
    return img.transform((w + shift, h), Image.AFFINE, (1, sh, -shift if sh > 0 else 0, 0, 1, 0),
                         resample=Image.BICUBIC, fillcolor=(255, 255, 255))


def ref_background(img, bg):                 # transformation.py:328-345
    bg_rgb = tuple(int(c * 255) for c in bg)
    background = Image.new('RGB', img.size, bg_rgb)
    gray = np.array(img.convert('RGB').convert('L'))
    edges = ndimage.sobel(gray)
    mask = binary_dilation(edges > np.percentile(edges, 70), iterations=3)
    return Image.composite(img.convert('RGB'), background, Image.fromarray((mask * 255).astype(np.uint8)))


def main():
    cases = {}
    meta = []

    def add(name, seed, hw, params, out, pinned=True):
        key = f"{name}__{len(meta):03d}"
        cases[key] = np.asarray(out)
        meta.append((key, name, seed, hw[0], hw[1], repr(params), int(pinned)))

    for i, hw in enumerate(SIZES):
        seed = 100 + i
        a = synth(seed, *hw)
        img = Image.fromarray(a)
        w, h = img.size
        for ang in (-22.5, -2.5, 0.0, 10.0, 22.5, 30.0, 90.0, 180.0):       # transformation.py:200
            add("rotation", seed, hw, ang, img.rotate(-ang, fillcolor=(0, 0, 0), expand=False))
        add("rotate_bilinear", seed, hw, 30.0, img.rotate(30.0, resample=Image.BILINEAR, fillcolor=(0, 0, 0)))
        m = O.rotate_zoom_matrix(w, h, 30.0, 1.5)                             # benchmark configs[3]
        add("affine_bilinear_rot30_zoom1.5", seed, hw, m,
            img.transform((w, h), Image.AFFINE, m, resample=Image.BILINEAR, fillcolor=(0, 0, 0)))
        for s in O.grid_values("scale") + [1.5]:                               # transformation.py:179-194
            add("scale", seed, hw, s, ref_scale(img, s))
        for sh in (0.0, 0.30000000000000004, 1.0):
            add("shear", seed, hw, sh, ref_shear(img, sh))
        for b in (-0.05, 0.0, 0.05):                                           # transformation.py:266-267
            add("brightness", seed, hw, b, ImageEnhance.Brightness(img).enhance(1.0 + b))
        for t in ((0, 0), (5, -10), (-50, 50), (45, 45)):
            add("translation", seed, hw, t, ref_translation(img, *t))
        gray = np.array(img.convert('L'))                                      # transformation.py:336
        add("rgb2l", seed, hw, None, gray)
        add("sobel_x_wrap", seed, hw, None, ndimage.sobel(gray))               # transformation.py:339
        add("background_change", seed, hw, (0.2, 0.5, 0.9), ref_background(img, (0.2, 0.5, 0.9)))
        bg = Image.new('RGB', img.size, (51, 127, 229))
        add("background_simple", seed, hw, (0.2, 0.5, 0.9), Image.blend(img, bg, 0.3))   # :354
        rng = np.random.RandomState(7)
        noise = rng.normal(0, 0.05 * 255, a.shape).astype(np.float32)         # transformation.py:275-278
        add("gaussian_noise_seed7_std0.05", seed, hw, 0.05,
            np.clip(a.astype(np.float32) + noise, 0, 255).astype(np.uint8))
        # ---- cv2-backed: float definitions only (parity unpinned)
        for r in (0.5, 5 / 6, 1.0, 2.5):
            add("blur", seed, hw, r, O.apply_blur(a, r), pinned=False)
        for al in (0.0, 0.30000000000000004, 1.0):
            add("contrast", seed, hw, al, O.apply_contrast(a, al), pinned=False)
        add("motion_blur", seed, hw, 5, O.motion_blur(a, 5), pinned=False)
        add("sobel_magnitude", seed, hw, None, O.sobel_magnitude(gray), pinned=False)   # no reference counterpart
        add("gray_box3", seed, hw, None, O.gray_box3(a), pinned=False)

    np.savez_compressed(os.path.join(HERE, "hotpath_golden.npz"), **cases)
    with open(os.path.join(HERE, "hotpath_golden_index.tsv"), "w") as fh:
        fh.write("key\top\tseed\th\tw\tparams\tpinned\n")
        for row in meta:
            fh.write("\t".join(map(str, row)) + "\n")

    # sha256 of full-size outputs of the integer-exact ops (too big to commit as arrays)
    with open(os.path.join(HERE, "fullsize_sha256.tsv"), "w") as fh:
        fh.write("op\th\tw\tparams\tsha256\n")
        for (h, w) in ((1080, 1920), (2160, 3840)):
            a = synth(12345, h, w)
            img = Image.fromarray(a)
            rows = [
                ("rotation", 30.0, img.rotate(-30.0, fillcolor=(0, 0, 0), expand=False)),
                ("scale", 1.1, ref_scale(img, 1.1)),
                ("brightness", 0.05, ImageEnhance.Brightness(img).enhance(1.05)),
                ("rgb2l", None, img.convert('L')),
                ("sobel_x_wrap", None, ndimage.sobel(np.array(img.convert('L')))),
                ("translation", (45, -50), ref_translation(img, 45, -50)),
                ("affine_bilinear_rot30_zoom1.5", None,
                 img.transform((w, h), Image.AFFINE, O.rotate_zoom_matrix(w, h, 30.0, 1.5),
                               resample=Image.BILINEAR, fillcolor=(0, 0, 0))),
            ]
            for op, prm, out in rows:
                fh.write(f"{op}\t{h}\t{w}\t{prm!r}\t{hashlib.sha256(np.asarray(out).tobytes()).hexdigest()}\n")
    print(f"wrote {len(meta)} cases")
    driver_fixture()


DRIVER_ORDER = ['scale', 'rotation', 'lighten_darken', 'gaussian_noise', 'translation', 'contrast', 'blur', 'shear']


def driver_fixture():
    """for seed S and a synthetic 48x64 RGB input, the driver loop of
    /root/reference/transformation.py:113-139 (one `random.choice(grid)` per transform type, two for
    translation, in the dict's order; file name f"{name}_{type}_{value}_corrupted.jpg") with every
    transform evaluated by the LIBRARY call the reference makes -> (type, value, file name, sha256).
    Rows of the cv2-backed types (contrast, blur) carry the float-definition hash and pinned = 0."""
    import random
    a = synth(5, 48, 64)
    img = Image.fromarray(a)
    with open(os.path.join(HERE, "driver_fixture.tsv"), "w") as fh:
        fh.write("seed\ttype\tvalue\tfilename\tsha256\tpinned\n")
        for seed in (1234, 7, 2024):
            random.seed(seed)
            np.random.seed(99)
            for ttype in DRIVER_ORDER:
                grid = O.grid_values(ttype)
                pinned = 1
                if ttype == 'translation':
                    tx = random.choice(grid); ty = random.choice(grid)
                    value, fname = (tx, ty), f"img0_{ttype}_{tx}_{ty}_corrupted.jpg"
                    out = ref_translation(img, tx, ty)
                else:
                    v = random.choice(grid)
                    value, fname = (v,), f"img0_{ttype}_{v}_corrupted.jpg"
                    if ttype == 'scale': out = ref_scale(img, v)                                        # :173-196
                    elif ttype == 'rotation': out = img.rotate(-v, fillcolor=(0, 0, 0), expand=False)    # :198-201
                    elif ttype == 'lighten_darken': out = ImageEnhance.Brightness(img).enhance(1.0 + v)  # :261-269
                    elif ttype == 'gaussian_noise':                                                      # :272-281
                        noise = np.random.normal(0, v * 255, a.shape).astype(np.float32)
                        out = np.clip(a.astype(np.float32) + noise, 0, 255).astype(np.uint8)
                    elif ttype == 'shear': out = ref_shear(img, v)                                       # :212-226
                    elif ttype == 'contrast': out, pinned = O.apply_contrast(a, v), 0                    # cv2: unpinned
                    else: out, pinned = O.apply_blur(a, v), 0                                            # cv2: unpinned
                sha = hashlib.sha256(np.ascontiguousarray(np.asarray(out)).tobytes()).hexdigest()
                fh.write(f"{seed}\t{ttype}\t{value!r}\t{fname}\t{sha}\t{pinned}\n")


if __name__ == "__main__":
    main()
