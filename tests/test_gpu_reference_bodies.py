"""GPU: the drop-in facade (imagetransformations_amd.transformation.apply_*, HIP kernels through the C-ABI) against the
outputs of the REFERENCE'S OWN FUNCTION BODIES (tests/golden/reference_bodies.npz, made in the build container by
tests/golden/make_golden_from_reference.py from /root/reference/transformation.py:173-354).  Same call, same
arguments, same PIL types as a test of the reference itself would use; bit-exact."""
import ast
import csv
import os

import numpy as np
import pytest
from PIL import Image

from conftest import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
INDEX = list(csv.DictReader(open(os.path.join(GOLD, "reference_bodies_index.tsv")), delimiter="\t"))
DATA = np.load(os.path.join(GOLD, "reference_bodies.npz"))
FNS = sorted({r["fn"] for r in INDEX})


@pytest.mark.parametrize("fn", FNS)
def test_facade_equals_the_reference_function_bodies(device, fn):
    from imagetransformations_amd import transformation as T
    rows = [r for r in INDEX if r["fn"] == fn]
    assert rows
    for row in rows:
        a = synth(int(row["seed"]), int(row["h"]), int(row["w"]))
        prm = ast.literal_eval(row["params"])
        img = Image.fromarray(a)
        if fn == "apply_gaussian_noise":
            np.random.seed(prm[1])                           # the reference's global NumPy stream (:274)
            out = T.apply_gaussian_noise(img, prm[0])
        elif fn in ("apply_background_change", "apply_background_change_simple"):
            out = getattr(T, fn)(img, prm)
        else:
            out = getattr(T, fn)(img, *prm)
        assert isinstance(out, Image.Image) and np.array_equal(np.asarray(img), a)       # inputs are never mutated
        want = DATA[row["key"]]
        got = np.asarray(out)
        assert got.shape == want.shape and np.array_equal(got, want), row["key"]


NEXT_INDEX = list(csv.DictReader(open(os.path.join(GOLD, "reference_bodies_next_index.tsv")), delimiter="\t"))
NEXT = np.load(os.path.join(GOLD, "reference_bodies_next.npz"))
NEXT_FNS = sorted({r["fn"] for r in NEXT_INDEX})


@pytest.mark.parametrize("fn", NEXT_FNS)
def test_next_row_facades_equal_the_reference_function_bodies(device, fn):
    """TransformationPool's cv2-free members (pipenline/cifar_image_transformations.py:39-107), the AugMix
    operation set (fall_2025/AugMix.py:30-37) and vert_flip / rand_crop / apply_random_zoom
    (fall_2025/transformations_code:39-52) — outputs of the reference files themselves, bit for bit."""
    import random
    from imagetransformations_amd import augmix, transformations_code
    from imagetransformations_amd.pool import TransformationPool
    owner, member = fn.split(".")
    target = getattr({"pool": TransformationPool, "augmix": augmix, "tcode": transformations_code}[owner], member)
    for row in [r for r in NEXT_INDEX if r["fn"] == fn]:
        a = synth(int(row["seed"]), int(row["h"]), int(row["w"]))
        prm = ast.literal_eval(row["params"])
        img = Image.fromarray(a)
        if member in ("gaussian_noise", "impulse_noise", "shot_noise"):
            np.random.seed(prm[1]); out = target(img, prm[0])
        elif member == "rotate":
            random.seed(prm[1]); out = target(img, prm[0])
        elif member == "rand_crop":
            np.random.seed(prm[0]); out = target(img)
        else:
            out = target(img, *prm)
        want = NEXT[row["key"]]
        got = np.asarray(out)
        assert got.shape == want.shape and np.array_equal(got, want), row["key"]
