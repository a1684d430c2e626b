"""CPU: the NumPy oracle against the committed golden vectors (tests/golden, produced by
calling Pillow / SciPy with the reference's argument lists — see make_golden.py)."""
import ast
import csv
import os

import numpy as np
import pytest

from conftest import synth
from oracle import imgxf_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_index():
    with open(os.path.join(GOLD, "hotpath_golden_index.tsv")) as fh:
        return list(csv.DictReader(fh, delimiter="\t"))


INDEX = load_index()
DATA = np.load(os.path.join(GOLD, "hotpath_golden.npz"))


def oracle_eval(op, a, prm):
    h, w = a.shape[:2]
    if op == "rotation": return O.apply_rotation(a, prm)
    if op == "rotate_bilinear": return O.rotate_bilinear(a, prm)
    if op == "affine_bilinear_rot30_zoom1.5": return O.affine_bilinear(a, (w, h), prm, fill=(0, 0, 0))
    if op == "scale": return O.apply_scale(a, prm)
    if op == "shear": return O.apply_shear(a, prm)
    if op == "brightness": return O.apply_brightness(a, prm)
    if op == "translation": return O.apply_translation(a, *prm)
    if op == "rgb2l": return O.rgb2l(a)
    if op == "sobel_x_wrap": return O.sobel_scipy(O.rgb2l(a))
    if op == "background_change": return O.apply_background_change(a, prm)
    if op == "background_simple": return O.apply_background_change_simple(a, prm)
    if op == "gaussian_noise_seed7_std0.05":
        noise = np.random.RandomState(7).normal(0, prm * 255, a.shape).astype(np.float32)
        return O.add_noise(a, noise)
    if op == "blur": return O.apply_blur(a, prm)
    if op == "contrast": return O.apply_contrast(a, prm)
    if op == "motion_blur": return O.motion_blur(a, prm)
    if op == "sobel_magnitude": return O.sobel_magnitude(O.rgb2l(a))
    if op == "gray_box3": return O.gray_box3(a)
    raise KeyError(op)


@pytest.mark.parametrize("row", INDEX, ids=[r["key"] for r in INDEX])
def test_oracle_matches_golden(row):
    a = synth(int(row["seed"]), int(row["h"]), int(row["w"]))
    prm = ast.literal_eval(row["params"])
    got = oracle_eval(row["op"], a, prm)
    want = DATA[row["key"]]
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_golden_covers_pinned_and_unpinned_ops():
    ops = {r["op"]: int(r["pinned"]) for r in INDEX}
    # Pillow / SciPy / NumPy backed: pinned by the libraries themselves
    for op in ("rotation", "scale", "shear", "brightness", "translation", "rgb2l", "sobel_x_wrap",
               "background_change", "affine_bilinear_rot30_zoom1.5"):
        assert ops[op] == 1
    # OpenCV backed: parity unpinned (no cv2 in this environment)
    for op in ("blur", "contrast", "motion_blur"):
        assert ops[op] == 0


def test_grid_values_reproduce_the_reference_float_artefacts():
    """transformation.py:126-127: min + j*step, int((max-min)/step)+1 values."""
    assert O.grid_values("scale") == [0.9, 1.0, 1.1, 1.2000000000000002, 1.3]   # 1.4 is never reached
    assert O.grid_values("shear")[3] == 0.30000000000000004
    assert len(O.grid_values("rotation")) == 19 and O.grid_values("rotation")[9] == 0.0
    assert len(O.grid_values("blur")) == 11 and O.grid_values("blur")[-1] == 5.0
    assert [O.blur_ksize(r) for r in O.grid_values("blur")] == [None, 3, 7, 9, 13, 15, 19, 21, 25, 27, 31]
    assert O.blur_ksize(5 / 6) == 5      # the benchmark's "5x5" under the reference's own rule


def test_oracle_reproduces_the_driver_fixture():
    """SURVEY §8a row H on the CPU: the oracle evaluated on the fixture's drawn values gives the
    bytes Pillow / NumPy gave (tests/golden/driver_fixture.tsv); grids and draw order are the
    reference's (one random.choice per type, two for translation)."""
    import hashlib
    import random
    rows = list(csv.DictReader(open(os.path.join(GOLD, "driver_fixture.tsv")), delimiter="\t"))
    assert len(rows) == 24
    a = synth(5, 48, 64)
    for seed in sorted({int(r["seed"]) for r in rows}):
        random.seed(seed)
        np.random.seed(99)
        for r in [r for r in rows if int(r["seed"]) == seed]:
            ttype, grid = r["type"], O.grid_values(r["type"])
            if ttype == "translation":
                args = (random.choice(grid), random.choice(grid))
            else:
                args = (random.choice(grid),)
            assert repr(args) == r["value"]
            if ttype == "scale": out = O.apply_scale(a, *args)
            elif ttype == "rotation": out = O.apply_rotation(a, *args)
            elif ttype == "lighten_darken": out = O.apply_brightness(a, *args)
            elif ttype == "gaussian_noise": out = O.apply_gaussian_noise(a, *args)
            elif ttype == "translation": out = O.apply_translation(a, *args)
            elif ttype == "shear": out = O.apply_shear(a, *args)
            elif ttype == "contrast": out = O.apply_contrast(a, *args)
            else: out = O.apply_blur(a, *args)
            assert hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest() == r["sha256"], (seed, ttype)
