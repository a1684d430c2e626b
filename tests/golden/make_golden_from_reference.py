#!/usr/bin/env python3
"""Golden vectors produced by the REFERENCE'S OWN FUNCTION BODIES.

Build container only (needs /root/reference, Pillow, SciPy, pandas, matplotlib):

    python tests/golden/make_golden_from_reference.py

`/root/reference/transformation.py` cannot be imported as it stands: line 7 is `import cv2` and OpenCV is
not installed; lines 16-17 create directories under /Users/... at import time.  This script

  * registers an EMPTY module object under the name `cv2` (it has no attributes: any reference function that
    touches cv2 raises AttributeError, so nothing cv2-backed can be recorded by accident — `apply_contrast`,
    `apply_blur` stay "parity unpinned", their arithmetic is not stubbed anywhere),
  * turns `os.makedirs` into a no-op for the duration of the import only,
  * loads the file with importlib from where it lies (nothing of it is copied into this repository),

and records, for synthetic inputs `default_rng(seed).integers(0, 256)`, the outputs of every function of
transformation.py:173-354 that never reaches cv2:

    apply_scale, apply_rotation, apply_shear, apply_brightness, apply_gaussian_noise (np.random seeded),
    apply_translation, apply_camera_distance, apply_xy_translation_3d, apply_rotation_3d,
    apply_background_change, apply_background_change_simple

on the reference's own parameter grids (:95-105) into `reference_bodies.npz` + `reference_bodies_index.tsv`.
It also runs the reference's driver loop `apply_all_transformations` (:92-170) itself, with the two cv2-backed
members replaced by pass-throughs WHOSE OUTPUTS ARE NOT RECORDED, to pin the `random` draw order, the
parameter values and the file names of all eight types plus the pixels of the six cv2-free ones
(`reference_driver_fixture.tsv`).  tests/test_oracle_golden.py asserts that oracle/imgxf_oracle.py reproduces
every recorded output bit for bit; only the data (inputs by seed, outputs) travel, never reference source.
The same is done for the files of SURVEY 8f (`record_next_rows`): the eight cv2-free TransformationPool members, the
AugMix operation set and vert_flip / rand_crop / apply_random_zoom -> `reference_bodies_next.npz`; their import lines
also name torchvision (not installed), which gets the same empty-module treatment — `apply_perspective_warp`, `augmix()`
and everything else that USES torchvision or cv2 is not recorded.
"""
import hashlib
import importlib.util
import os
import random
import sys
import tempfile
import types

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/transformation.py"
REF_POOL = "/root/reference/pipenline/cifar_image_transformations.py"
REF_AUGMIX = "/root/reference/fall_2025/AugMix.py"
REF_TCODE = "/root/reference/fall_2025/transformations_code"
SIZES = [(32, 32), (37, 61), (48, 64)]


def synth(seed, h, w):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


def load_reference(path=REF, name="reference_transformation"):
    """Import one reference file from where it lies.  `cv2` and `torchvision` (+ the two submodules the files name
    in their import lines) are EMPTY module objects: the import statements succeed, any use raises AttributeError."""
    for missing in ("cv2", "torchvision", "torchvision.transforms", "torchvision.transforms.functional"):
        have = sys.modules.get(missing)
        if have is not None and getattr(have, "__file__", None):
            raise SystemExit(f"a real {missing} is importable here: pin its rows with it instead of this script")
        sys.modules[missing] = types.ModuleType(missing)   # empty: satisfies the import line, provides nothing
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    real_makedirs = os.makedirs
    os.makedirs = lambda *a, **k: None                      # transformation.py:16-17 (module-level /Users/... paths)
    try:
        from importlib.machinery import SourceFileLoader     # `transformations_code` has no .py suffix
        loader = SourceFileLoader(name, path)
        spec = importlib.util.spec_from_loader(name, loader)
        mod = importlib.util.module_from_spec(spec)
        loader.exec_module(mod)
    finally:
        os.makedirs = real_makedirs
    return mod


def record_next_rows():
    """the cv2-free members of TransformationPool (pipenline/cifar_image_transformations.py:39-107),
    the AugMix operation set (fall_2025/AugMix.py:30-37) and vert_flip / rand_crop / apply_random_zoom
    (fall_2025/transformations_code:39-52), executed from the reference files -> reference_bodies_next.npz."""
    P = load_reference(REF_POOL, "reference_pool").TransformationPool
    A = load_reference(REF_AUGMIX, "reference_augmix")
    C = load_reference(REF_TCODE, "reference_tcode")
    cases, rows = {}, []

    def add(fn, seed, hw, params, out):
        key = f"{fn}__s{seed}_{hw[0]}x{hw[1]}__{'_'.join(repr(p) for p in params)}"
        cases[key] = np.asarray(out)
        rows.append((key, fn, seed, hw[0], hw[1], repr(tuple(params))))

    for si, hw in enumerate([(32, 32), (37, 61), (48, 64)]):
        seed = 400 + si
        img = Image.fromarray(synth(seed, *hw))
        for sev in (1, 2, 3, 4, 5):
            for fn in ("gaussian_noise", "impulse_noise", "shot_noise"):
                np.random.seed(2000 + seed + sev)             # the members draw from NumPy's global stream
                add(f"pool.{fn}", seed, hw, (sev, 2000 + seed + sev), getattr(P, fn)(img, sev))
            add("pool.defocus_blur", seed, hw, (sev,), P.defocus_blur(img, sev))
        for fac in (0.5, 0.8, 1.0, 1.3, 2.0, 3.0):
            for fn in ("enhance_contrast", "enhance_brightness", "enhance_sharpness", "enhance_color"):
                add(f"pool.{fn}", seed, hw, (fac,), getattr(P, fn)(img, fac))
        for sev in (1, 3, 5):
            for fn in ("posterize", "shear_x", "shear_y", "translate_x", "translate_y", "equalize", "solarize"):
                add(f"augmix.{fn}", seed, hw, (sev,), getattr(A, fn)(img, sev))
            random.seed(3000 + seed + sev)                    # rotate draws its sign from `random`
            add("augmix.rotate", seed, hw, (sev, 3000 + seed + sev), A.rotate(img, sev))
        add("tcode.vert_flip", seed, hw, (), C.vert_flip(img))
        for k in range(3 if hw[0] - int(0.78 * hw[1]) + 1 > 0 else 0):   # the reference raises when h < int(0.78 w)
            np.random.seed(4000 + seed + k)
            add("tcode.rand_crop", seed, hw, (4000 + seed + k,), C.rand_crop(img))
        for z in (1.0, 1.05, 1.1):
            add("tcode.apply_random_zoom", seed, hw, (z,), C.apply_random_zoom(img, z))
    np.savez_compressed(os.path.join(HERE, "reference_bodies_next.npz"), **cases)
    with open(os.path.join(HERE, "reference_bodies_next_index.tsv"), "w") as fh:
        fh.write("key\tfn\tseed\th\tw\tparams\n")
        for r in rows:
            fh.write("\t".join(str(v) for v in r) + "\n")
    return len(cases)


def grid(params):
    n = int((params["max"] - params["min"]) / params["step"]) + 1
    return [params["min"] + j * params["step"] for j in range(n)]


GRIDS = {                                                   # transformation.py:95-105 (values, not code)
    "scale": {"min": 0.9, "max": 1.4, "step": 0.1}, "rotation": {"min": -22.5, "max": 22.5, "step": 2.5},
    "lighten_darken": {"min": -0.05, "max": 0.05, "step": 0.01}, "gaussian_noise": {"min": 0.0, "max": 0.1, "step": 0.01},
    "translation": {"min": -50, "max": 50, "step": 5}, "shear": {"min": 0, "max": 1, "step": 0.1},
}


def main():
    R = load_reference()
    cases, rows = {}, []

    def add(fn, seed, hw, params, out):
        key = f"{fn}__s{seed}_{hw[0]}x{hw[1]}__{'_'.join(repr(p) for p in params)}"
        cases[key] = np.asarray(out)
        rows.append((key, fn, seed, hw[0], hw[1], repr(tuple(params))))

    for si, hw in enumerate(SIZES):
        seed = 300 + si
        a = synth(seed, *hw)
        img = Image.fromarray(a)
        for s in grid(GRIDS["scale"]) + [0.5, 1.5]:
            add("apply_scale", seed, hw, (s,), R.apply_scale(img, s))
        for ang in grid(GRIDS["rotation"]) + [30.0, 90.0, 180.0, 270.0, 360.0]:
            add("apply_rotation", seed, hw, (ang,), R.apply_rotation(img, ang))
            if ang in (-22.5, 7.5, 30.0):
                add("apply_rotation_3d", seed, hw, (ang,), R.apply_rotation_3d(img, ang))
        for sh in grid(GRIDS["shear"]):
            add("apply_shear", seed, hw, (sh,), R.apply_shear(img, sh))
        for b in grid(GRIDS["lighten_darken"]):
            add("apply_brightness", seed, hw, (b,), R.apply_brightness(img, b))
        for std in grid(GRIDS["gaussian_noise"])[::2]:
            np.random.seed(1000 + seed)                       # the reference draws from NumPy's global stream (:274)
            add("apply_gaussian_noise", seed, hw, (std, 1000 + seed), R.apply_gaussian_noise(img, std))
        for tx, ty in [(-50, 50), (-5, 0), (0, 0), (10, -15), (35, 45), (50, -50), (7.9, -3.2), (100, 3)]:
            add("apply_translation", seed, hw, (tx, ty), R.apply_translation(img, tx, ty))
        for d in (2.0, 2.75, 3.0, 2.5):
            add("apply_camera_distance", seed, hw, (d,), R.apply_camera_distance(img, d))
        for tx, ty in [(0.1, -0.2), (-0.25, 0.05), (0.0, 0.0)]:
            add("apply_xy_translation_3d", seed, hw, (tx, ty), R.apply_xy_translation_3d(img, tx, ty))
        for bg in [(0.2, 0.4, 0.9), (1.0, 1.0, 1.0), (0.0, 0.5, 0.0)]:
            add("apply_background_change", seed, hw, bg, R.apply_background_change(img, bg))
            add("apply_background_change_simple", seed, hw, bg, R.apply_background_change_simple(img, bg))

    np.savez_compressed(os.path.join(HERE, "reference_bodies.npz"), **cases)
    with open(os.path.join(HERE, "reference_bodies_index.tsv"), "w") as fh:
        fh.write("key\tfn\tseed\th\tw\tparams\n")
        for r in rows:
            fh.write("\t".join(str(v) for v in r) + "\n")

    # ---- the reference's own driver loop (:92-170): draws, values, file names (all 8 types), pixels (6 types)
    cv2_backed = ("contrast", "blur")
    R.apply_contrast = lambda img, v: img                    # pass-throughs so that the loop can run; NOT recorded
    R.apply_blur = lambda img, v: img
    drows = []
    with tempfile.TemporaryDirectory() as tmp:
        R.output_dir = tmp
        for seed in (1, 2, 3):
            hw = (48, 64)
            imgs = [(Image.fromarray(synth(500 + 10 * seed + j, *hw)), f"/data/img_{seed}_{j}.JPEG") for j in range(2)]
            random.seed(seed)
            np.random.seed(seed)
            outs = R.apply_all_transformations(imgs)
            names = sorted(os.listdir(tmp))
            assert len(outs) == 16
            # the loop does not return the names: recover them per (image, type) from the files it wrote
            k = 0
            for j in range(2):
                for t in ("scale", "rotation", "lighten_darken", "gaussian_noise", "translation", "contrast", "blur", "shear"):
                    mine = [n for n in names if n.startswith(f"img_{seed}_{j}_{t}_")]
                    assert len(mine) == 1, (t, mine)
                    digest = "-" if t in cv2_backed else hashlib.sha256(
                        np.asarray(outs[k]).tobytes() + repr(np.asarray(outs[k]).shape).encode()).hexdigest()
                    drows.append((seed, j, 500 + 10 * seed + j, t, mine[0], digest))
                    k += 1
            for n in names:
                os.remove(os.path.join(tmp, n))
    with open(os.path.join(HERE, "reference_driver_fixture.tsv"), "w") as fh:
        fh.write("seed\timage\tinput_seed\ttype\tfilename\tsha256\n")
        for r in drows:
            fh.write("\t".join(str(v) for v in r) + "\n")
    n_next = record_next_rows()
    print(f"{n_next} cases of the 8f files (Pool / AugMix ops / transformations_code)")
    print(f"{len(cases)} function cases, {len(drows)} driver rows "
          f"({sum(1 for r in drows if r[5] != '-')} with pixels) written to {HERE}")


if __name__ == "__main__":
    main()
