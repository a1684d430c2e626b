"""CPU: the NumPy oracle against outputs of the REFERENCE'S OWN FUNCTION BODIES
(/root/reference/transformation.py:173-354 executed in the build container by
tests/golden/make_golden_from_reference.py; only inputs-by-seed and outputs are committed).  Every function of
the file that never reaches cv2 is pinned here bit for bit; `apply_contrast` / `apply_blur` (cv2-backed) are
not and stay "parity unpinned".  The driver rows pin the reference loop's draw order, values and file names
for all eight types and the pixels of the six cv2-free ones."""
import ast
import csv
import hashlib
import os
import random

import numpy as np
import pytest

from conftest import synth
from oracle import imgxf_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
INDEX = list(csv.DictReader(open(os.path.join(GOLD, "reference_bodies_index.tsv")), delimiter="\t"))
DATA = np.load(os.path.join(GOLD, "reference_bodies.npz"))


def oracle_eval(fn, a, prm):
    if fn == "apply_gaussian_noise":
        std, seed = prm
        np.random.seed(seed)                       # the reference draws from NumPy's global stream (:274)
        return O.apply_gaussian_noise(a, std)
    if fn == "apply_rotation_3d":                  # transformation.py:324-325: a plain alias
        return O.apply_rotation(a, *prm)
    if fn in ("apply_background_change", "apply_background_change_simple"):
        return getattr(O, fn)(a, prm)
    return getattr(O, fn)(a, *prm)


@pytest.mark.parametrize("row", INDEX, ids=[r["key"] for r in INDEX])
def test_oracle_equals_the_reference_function_bodies(row):
    a = synth(int(row["seed"]), int(row["h"]), int(row["w"]))
    got = oracle_eval(row["fn"], a, ast.literal_eval(row["params"]))
    want = DATA[row["key"]]
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.array_equal(got, want)


def test_every_cv2_free_function_of_the_reference_is_pinned():
    fns = {r["fn"] for r in INDEX}
    assert fns == {"apply_scale", "apply_rotation", "apply_rotation_3d", "apply_shear", "apply_brightness",
                   "apply_gaussian_noise", "apply_translation", "apply_camera_distance", "apply_xy_translation_3d",
                   "apply_background_change", "apply_background_change_simple"}
    assert len(INDEX) == len(DATA.files) >= 200


def test_oracle_driver_equals_the_reference_loop():
    """The reference's `apply_all_transformations` (:92-170) run on seeded inputs: same `random` draws in the same
    order, same values (float artefacts included), same file names for all eight types; same pixels for the six
    types that do not go through cv2 (the noise draws come from the global NumPy stream, in loop order)."""
    rows = list(csv.DictReader(open(os.path.join(GOLD, "reference_driver_fixture.tsv")), delimiter="\t"))
    assert len(rows) == 48
    order = ["scale", "rotation", "lighten_darken", "gaussian_noise", "translation", "contrast", "blur", "shear"]
    for seed in (1, 2, 3):
        random.seed(seed)
        np.random.seed(seed)
        mine = [r for r in rows if int(r["seed"]) == seed]
        assert [r["type"] for r in mine] == order * 2
        for r in mine:
            a = synth(int(r["input_seed"]), 48, 64)
            grid = O.grid_values(r["type"])
            name = f"img_{seed}_{r['image']}"
            if r["type"] == "translation":
                args = (random.choice(grid), random.choice(grid))
                fname = f"{name}_translation_{args[0]}_{args[1]}_corrupted.jpg"
            else:
                args = (random.choice(grid),)
                fname = f"{name}_{r['type']}_{args[0]}_corrupted.jpg"
            assert fname == r["filename"]
            if r["sha256"] == "-":
                continue                                       # cv2-backed: draw and name only
            fn = {"scale": O.apply_scale, "rotation": O.apply_rotation, "lighten_darken": O.apply_brightness,
                  "gaussian_noise": O.apply_gaussian_noise, "translation": O.apply_translation, "shear": O.apply_shear}[r["type"]]
            out = np.ascontiguousarray(fn(a, *args))
            assert hashlib.sha256(out.tobytes() + repr(out.shape).encode()).hexdigest() == r["sha256"], (seed, r["type"], args)


NEXT_INDEX = list(csv.DictReader(open(os.path.join(GOLD, "reference_bodies_next_index.tsv")), delimiter="\t"))
NEXT = np.load(os.path.join(GOLD, "reference_bodies_next.npz"))


def next_oracle(fn, a, prm):
    """The oracle's restatement of a SURVEY 8f member, or None where the member is plain NumPy on the host in the
    product as well (the pool's noise draws) and has no oracle function of its own."""
    h, w = a.shape[:2]
    if fn == "pool.defocus_blur": return O.pil_gaussian_blur(a, [3, 4, 6, 8, 10][prm[0] - 1])
    if fn == "pool.enhance_contrast": return O.enhance_contrast(a, prm[0])
    if fn == "pool.enhance_brightness": return O.blend(np.zeros_like(a), a, prm[0])      # ImageEnhance.Brightness = blend with black
    if fn == "pool.enhance_sharpness": return O.enhance_sharpness(a, prm[0])
    if fn == "pool.enhance_color": return O.enhance_color(a, prm[0])
    if fn == "augmix.posterize": return O.posterize(a, int(prm[0]))
    if fn == "augmix.solarize": return O.solarize(a, int(prm[0] * 20))
    if fn == "augmix.equalize": return O.equalize(a)
    if fn == "augmix.shear_x": return O.affine_nearest(a, (w, h), (1, prm[0] * 0.3, 0, 0, 1, 0))
    if fn == "augmix.shear_y": return O.affine_nearest(a, (w, h), (1, 0, 0, prm[0] * 0.3, 1, 0))
    if fn == "augmix.translate_x": return O.affine_nearest(a, (w, h), (1, 0, prm[0] * 2, 0, 1, 0))
    if fn == "augmix.translate_y": return O.affine_nearest(a, (w, h), (1, 0, 0, 0, 1, prm[0] * 2))
    if fn == "augmix.rotate":
        random.seed(prm[1])
        return O.apply_rotation(a, -(prm[0] * random.choice([-1, 1])))    # img.rotate(x) = apply_rotation(-x), fill 0
    if fn == "tcode.vert_flip": return O.vert_flip(a)
    if fn == "tcode.rand_crop":
        np.random.seed(prm[0])
        cs = int(0.78 * w)
        x, y = np.random.randint(0, w - cs + 1), np.random.randint(0, h - cs + 1)
        return O.rand_crop(a, x, y)
    if fn == "tcode.apply_random_zoom": return O.apply_scale(a, prm[0])
    return None


@pytest.mark.parametrize("fn", sorted({r["fn"] for r in NEXT_INDEX}))
def test_oracle_equals_the_reference_bodies_of_the_next_rows(fn):
    """SURVEY 8f members executed from the reference files (make_golden_from_reference.record_next_rows)."""
    checked = 0
    for row in [r for r in NEXT_INDEX if r["fn"] == fn]:
        a = synth(int(row["seed"]), int(row["h"]), int(row["w"]))
        got = next_oracle(fn, a, ast.literal_eval(row["params"]))
        if got is None:
            continue
        want = NEXT[row["key"]]
        assert got.shape == want.shape and np.array_equal(got, want), row["key"]
        checked += 1
    assert checked or fn in ("pool.gaussian_noise", "pool.impulse_noise", "pool.shot_noise")
