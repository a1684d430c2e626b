"""Drop-in for the reference's `transformation.py` per-image functions, backed by HIP.

Same names, argument meaning, return types and error behaviour as
/root/reference/transformation.py:173-354 (`apply_<x>(img: PIL.Image, ...) -> PIL.Image`)
and its driver `apply_all_transformations` (:92-170).  Each body is
    PIL image -> uint8 HWC device tensor -> libimgxf kernel(s) -> PIL image
and fails loudly when the HIP library or a ROCm device is missing (no CPU path).
Batches of frames should use `imagetransformations_amd.ops` directly and stay on the device.
"""
from __future__ import annotations

import math
import os
import random
from typing import List, Tuple

import numpy as np
import torch
from PIL import Image

from . import ops, staging

# Where apply_all_transformations writes its JPEGs (the reference hard-codes a
# /Users/... path at transformation.py:13-17).  None = do not write files.
output_dir: str | None = os.environ.get("IMGXF_OUTPUT_DIR")

TRANSFORMATIONS_2D = {                      # transformation.py:95-105
    'scale': {'min': 0.9, 'max': 1.4, 'step': 0.1},
    'rotation': {'min': -22.5, 'max': 22.5, 'step': 2.5},
    'lighten_darken': {'min': -0.05, 'max': 0.05, 'step': 0.01},
    'gaussian_noise': {'min': 0.0, 'max': 0.1, 'step': 0.01},
    'translation': {'min': -50, 'max': 50, 'step': 5},
    'contrast': {'min': 0, 'max': 1, 'step': 0.1},
    'blur': {'min': 0, 'max': 5, 'step': 0.5},
    'shear': {'min': 0, 'max': 1, 'step': 0.1},
}


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("imagetransformations_amd needs a ROCm device (MI355X); none is visible "
                           "and there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _upload(img: Image.Image) -> torch.Tensor:
    """np.array(img) (HWC / HW uint8) -> device tensor."""
    arr = np.array(img)
    if arr.dtype != np.uint8:
        raise TypeError(f"only 8-bit images are supported, got mode {img.mode!r}")
    return staging.upload(arr, _device())


def _download(t: torch.Tensor) -> Image.Image:
    """Image.fromarray(device tensor): asynchronous copy into pinned memory, wait for that copy only.  An RGB frame goes
    back as RGBX and becomes a PIL image mapped onto the pinned block (staging.image_from_rgbx) while its budget lasts."""
    if t.dim() == 3 and t.shape[-1] == 3 and t.dtype == torch.uint8 and staging.zero_copy_reserve(t.shape[0] * t.shape[1] * 4):
        return staging.image_from_rgbx(staging.download(ops.permute_channels(t, (0, 1, 2, 2))).numpy())
    return Image.fromarray(staging.download(t).numpy())


def _hw(t: torch.Tensor):
    """(height, width) of a [H,W], [H,W,C] or [N,H,W,C] tensor."""
    return (t.shape[0], t.shape[1]) if t.dim() == 2 else (t.shape[-3], t.shape[-2])


def _drop_alpha(t: torch.Tensor) -> torch.Tensor:
    return ops.permute_channels(t, (0, 1, 2)) if t.dim() == 3 and t.shape[2] == 4 else t


# ------------------------------------------------------------------ scale (:173-196)
def _scale_t(t: torch.Tensor, scale_factor: float) -> torch.Tensor:
    """apply_scale on a [H,W,C] or [N,H,W,C] device tensor."""
    height, width = _hw(t)
    new_width = int(width * scale_factor)
    new_height = int(height * scale_factor)
    if scale_factor > 1.0:
        # resize + centre crop in one pass over the pixels that survive the crop
        left = (new_width - width) // 2
        top = (new_height - height) // 2
        return ops.resize_crop(t, (new_width, new_height), (left, top, left + width, top + height))
    if scale_factor < 1.0 and new_width >= 1 and new_height >= 1:
        # resize straight into its place on the black canvas (the paste at :188-194)
        canvas = ops.new(t, height, width, (0, 0, 0))
        left, top = (width - new_width) // 2, (height - new_height) // 2
        window = canvas[..., top:top + new_height, left:left + new_width, :] if t.dim() >= 3 else \
            canvas[top:top + new_height, left:left + new_width]
        ops.resize(t, (new_width, new_height), ops.RESAMPLE_LANCZOS, out=window)
        return canvas
    return ops.resize_lanczos(t, (new_width, new_height))


def apply_scale(img: Image.Image, scale_factor: float) -> Image.Image:
    return _download(_scale_t(_upload(img), scale_factor))


# ------------------------------------------------------------------ rotation (:198-201)
def _rotation_t(t: torch.Tensor, angle: float) -> torch.Tensor:
    return ops.rotate(t, -angle, ops.NEAREST, fillcolor=(0, 0, 0))


def apply_rotation(img: Image.Image, angle: float) -> Image.Image:
    return _download(_rotation_t(_upload(img), angle))


# ------------------------------------------------------------------ contrast (:203-210)
def apply_contrast(img: Image.Image, contrast_amount: float) -> Image.Image:
    t = _upload(img)
    if t.dim() == 2:
        raise IndexError("tuple index out of range")   # img_np.shape[2] on a 2-D array (:205)
    return _download(ops.scale_abs(_drop_alpha(t), contrast_amount, 0.0))


# ------------------------------------------------------------------ shear (:212-226)
def _shear_t(t: torch.Tensor, shear_factor: float) -> torch.Tensor:
    height, width = _hw(t)
    shift_in_pixels = int(math.ceil(shear_factor * height))
    matrix = (1, shear_factor, -shift_in_pixels if shear_factor > 0 else 0, 0, 1, 0)
    return ops.affine(t, matrix, (width + shift_in_pixels, height), ops.BICUBIC, fillcolor=(255, 255, 255))


def apply_shear(img: Image.Image, shear_factor: float) -> Image.Image:
    return _download(_shear_t(_upload(img), shear_factor))


# ------------------------------------------------------------------ blur (:228-257)
def _blur_ksize(blur_radius: float) -> int:
    """The reference's kernel-size rule (:239-246); 0 means "return the input object"."""
    ksize = int(blur_radius * 6)
    if ksize % 2 == 0:
        ksize += 1
    if ksize < 3 and blur_radius > 0:
        ksize = 3
    elif blur_radius == 0:
        return 0
    return ksize


# False (default): cv2.GaussianBlur by its float definition, the <= 1e-5 contract of BASELINE.json.  True (or
# IMGXF_BLUR_FIXED_POINT=1): OpenCV's uint8 fixed-point evaluation (restated, unpinned), which the reference's own output
# files sit marginally closer to (README.md)
BLUR_FIXED_POINT = os.environ.get("IMGXF_BLUR_FIXED_POINT", "0") == "1"


def apply_blur(img: Image.Image, blur_radius: float) -> Image.Image:
    """cv2.GaussianBlur by its float definition (the contract, to 1e-5 relative).  With the
    module switch BLUR_FIXED_POINT the 8-bit fixed-point evaluation of OpenCV >= 4 is used instead
    (restated and unpinned, but closer to the reference's own outputs)."""
    ksize = _blur_ksize(blur_radius)
    if ksize == 0:
        return img          # the reference hands back the input object itself (:245-246)
    t = _drop_alpha(_upload(img))
    return _download(ops.gaussian_blur(t, ksize, blur_radius, fixed_point=BLUR_FIXED_POINT))


# ------------------------------------------------------------------ brightness (:261-269)
def apply_brightness(img: Image.Image, brightness_factor: float) -> Image.Image:
    if img.mode not in ("RGB", "L"):
        raise NotImplementedError(f"apply_brightness supports RGB and L images, got {img.mode!r}")
    return _download(ops.brightness(_upload(img), 1.0 + brightness_factor))


# ------------------------------------------------------------------ gaussian noise (:272-281)
# "numpy" (default): the reference's own stream, bit-exact parity for the same np.random.seed.  Since late round 3 the
#   numbers themselves are computed on the device (numpy_stream.draw_on_device: MT19937 blocks + legacy_gauss restated, the
#   global generator's state advanced as np.random.normal would have; host draw if a sample is too close to a float32
#   rounding boundary for the device's log to be trusted).  "numpy-host" (IMGXF_NOISE_RNG=numpy-host): np.random.normal
#   on the host, as the reference itself.
# "device" (IMGXF_NOISE_RNG=device, opt-in): Philox4x32-10 + Box-Muller inside the add kernel; ONE np.random draw
# per call supplies the seed (so np.random.seed still makes a run repeatable), the pixels differ from the reference's
# for the same seed: distribution-level parity only (tests/test_gpu_noise_rng.py).  The host draw
# is ~8 of the 8.6 ms per 375 x 500 image in the batched driver.
NOISE_RNG = os.environ.get("IMGXF_NOISE_RNG", "numpy")
NOISE_DEVICE_MIN = 1 << 16                               # normals per call below which the host draws them itself


def _noise_seed() -> int:
    return int(np.random.randint(0, 2 ** 63 - 1, dtype=np.int64))


def apply_gaussian_noise(img: Image.Image, noise_std: float) -> Image.Image:
    """The noise is drawn on the host from NumPy's global generator exactly as the reference
    does (same stream for the same np.random.seed), then added and clipped on the device.
    With NOISE_RNG == "device" the normals are generated on the device instead (opt-in, see above)."""
    if NOISE_RNG == "device":
        return _download(ops.add_noise_device(_upload(img), noise_std * 255, _noise_seed()))
    img_array = np.array(img)
    dev = _device()
    z = _numpy_noise([(img_array.size, noise_std * 255)], dev)[0]
    if z is None:
        z = torch.from_numpy(np.random.normal(0, noise_std * 255, img_array.shape).astype(np.float32)).to(dev)
    out = ops.add_noise(torch.from_numpy(img_array).to(dev), z.view(img_array.shape))
    return _download(out)


def _numpy_noise(requests, dev):
    """[(count, scale)] -> float32 device tensors holding np.random.normal(0, scale, count).astype(float32) for each request,
    np.random's state advanced accordingly; [None, ...] (state untouched) when the host has to draw: mode "numpy-host", or
    numpy_stream's uncertainty guard."""
    if NOISE_RNG == "numpy-host" or not requests or sum(n for n, _ in requests) < NOISE_DEVICE_MIN:
        return [None] * len(requests)               # (a CIFAR image is 3072 normals: 50 us on the host, less than one launch)
    from . import numpy_stream
    state = np.random.get_state()
    try:
        got = numpy_stream.draw_on_device(requests, dev)
    except ValueError:                                  # fewer accepted groups than 12 standard deviations allow for: the host draws
        np.random.set_state(state)
        got = None
    return got if got is not None else [None] * len(requests)


# ------------------------------------------------------------------ translation (:284-307)
def _translation_t(t: torch.Tensor, tx: float, ty: float) -> torch.Tensor:
    """:284-307: black canvas, crop of what stays visible, paste at max(0, int(t)) — i.e. every pixel moves by
    (int(tx), int(ty)) and what moves in from outside is black; one kernel instead of fill + copy."""
    return ops.translate(t, int(tx), int(ty), (0, 0, 0))


def apply_translation(img: Image.Image, tx: float, ty: float) -> Image.Image:
    return _download(_translation_t(_upload(img.convert('RGB') if img.mode != 'RGB' else img), tx, ty))


def apply_camera_distance(img: Image.Image, distance_factor: float) -> Image.Image:   # :309-314
    return apply_scale(img, 2.75 / distance_factor)


def apply_xy_translation_3d(img: Image.Image, tx: float, ty: float) -> Image.Image:   # :316-321
    width, height = img.size
    return apply_translation(img, int(tx * width), int(ty * height))


def apply_rotation_3d(img: Image.Image, angle: float) -> Image.Image:                 # :324-325
    return apply_rotation(img, angle)


# ------------------------------------------------------------------ background (:328-354)
def _rgb_tensor(img: Image.Image) -> torch.Tensor:
    t = _upload(img)
    if t.dim() == 2:
        raise NotImplementedError("background change expects a colour image")
    return _drop_alpha(t)


def apply_background_change(img: Image.Image, bg_color: Tuple[float, float, float]) -> Image.Image:
    bg_rgb = tuple(int(c * 255) for c in bg_color)
    rgb = _rgb_tensor(img)
    edges = ops.rgb_sobel(rgb)                            # ndimage.sobel(img.convert('L')) (:336-339), L never stored
    edge_mask = ops.percentile_mask(edges, 70)            # edges > np.percentile(edges, 70)
    foreground = ops.dilate_cross(edge_mask, 3)           # binary_dilation(iterations=3)
    return _download(ops.composite_const(rgb, bg_rgb, foreground))   # Image.composite(img, Image.new(.., bg), mask)


def apply_background_change_simple(img: Image.Image, bg_color: Tuple[float, float, float]) -> Image.Image:
    bg_rgb = tuple(int(c * 255) for c in bg_color)
    return _download(ops.blend(_rgb_tensor(img), bg_rgb, 0.3))


# ------------------------------------------------------------------ driver (:73-170)
def load_data(data_path):
    """Walk `data_path`, open every *.jpeg as RGB -> [(PIL image, path)] (:73-89)."""
    image_paths = []
    for root, _, files in os.walk(data_path):
        image_paths.extend(os.path.join(root, f) for f in files if f.lower().endswith('.jpeg'))
    images = []
    for path in image_paths:
        try:
            images.append((Image.open(path).convert("RGB"), path))
        except Exception as e:      # the reference only guards the file loading
            print(f"Failed to load image {path}: {e}")
    print(f"Loaded {len(images)} images.")
    return images


_DISPATCH = {
    'scale': apply_scale,
    'rotation': apply_rotation,
    'lighten_darken': apply_brightness,
    'gaussian_noise': apply_gaussian_noise,
    'contrast': apply_contrast,
    'shear': apply_shear,
    'blur': apply_blur,
}


def grid_values(params) -> List[float]:
    """min + j*step for j < int((max-min)/step)+1 — float artefacts included (:126-127)."""
    num_steps = int((params['max'] - params['min']) / params['step']) + 1
    return [params['min'] + j * params['step'] for j in range(num_steps)]


def plan_transformations(name: str):
    """One random draw per transform type in the reference's order -> [(type, args, filename)].
    Consumes `random` exactly like the loop at transformation.py:119-139."""
    plan = []
    for transform_type, params in TRANSFORMATIONS_2D.items():
        values = grid_values(params)
        if transform_type == 'translation':
            tx = random.choice(values)
            ty = random.choice(values)
            plan.append((transform_type, (tx, ty), f"{name}_{transform_type}_{tx}_{ty}_corrupted.jpg"))
        else:
            value = random.choice(values)
            plan.append((transform_type, (value,), f"{name}_{transform_type}_{value}_corrupted.jpg"))
    return plan


JPEG_ON_DEVICE = os.environ.get("IMGXF_JPEG_DEVICE", "0") == "1"


def save_image(img: Image.Image, path: str) -> None:
    """The reference's `transformed.save(path)` (:161-162).  With `JPEG_ON_DEVICE` (or IMGXF_JPEG_DEVICE=1) an RGB image
    bound for a *.jpg / *.jpeg file is encoded by the GPU writer (`jpeg.encode`: the file Pillow would write, byte for
    byte); every other mode / format, and images carrying a comment Pillow would embed, go through Pillow."""
    if (JPEG_ON_DEVICE and img.mode == "RGB" and path.lower().endswith((".jpg", ".jpeg")) and "comment" not in img.info
            and min(img.size) > 0):
        from . import jpeg
        data = jpeg.encode(_upload(img)[None])[0]
        with open(path, "wb") as f:
            f.write(data)
    else:
        img.save(path)


DRIVER = os.environ.get("IMGXF_DRIVER", "batched")       # "per-image": apply_all_transformations runs the reference's literal loop
DRIVER_CHUNK = 256                                        # images per pass of the batched driver inside apply_all_transformations


def apply_all_transformations(images):
    """images: [(PIL image, path)] -> list of transformed PIL images (8 per input), /root/reference/transformation.py:92-170.
    Same draws (`random`, `np.random`), same names, same order, same pixels as the literal loop
    (`apply_all_transformations_per_image`, which tests/test_gpu_facade.py holds it against); since late round 3 the work is
    done DRIVER_CHUNK images at a time by the batched driver — images of one size share their uploads, launches and copies
    back.  A chunk with an image that is not 8-bit RGB, and `IMGXF_DRIVER=per-image`, take the literal loop."""
    if DRIVER == "per-image":
        return apply_all_transformations_per_image(images)
    images = list(images)
    transformed_images = []
    for c0 in range(0, len(images), DRIVER_CHUNK):
        chunk = images[c0:c0 + DRIVER_CHUNK]
        if all(_is_rgb(img) for img, _ in chunk):
            transformed_images.extend(apply_all_transformations_batched(chunk))
        else:
            transformed_images.extend(apply_all_transformations_per_image(chunk, _progress=False))
        done = c0 + len(chunk)
        for mark in range((c0 // 1000 + 1) * 1000, done + 1, 1000):     # the reference's progress line, per thousand images
            print(f"Processed {mark}/{len(images)} original images, created {8 * mark} transformed images")
    return transformed_images


def apply_all_transformations_per_image(images, _progress: bool = True):
    """The reference's loop, call by call: one upload, one launch and one copy back per transformation."""
    transformed_images = []
    total_transforms = 0
    for i, (img, path) in enumerate(images):
        name = os.path.splitext(os.path.basename(path))[0]
        for transform_type, args, new_filename in plan_transformations(name):
            fn = apply_translation if transform_type == 'translation' else _DISPATCH[transform_type]
            transformed_img = fn(img, *args)
            if output_dir is not None:
                save_image(transformed_img, os.path.join(output_dir, new_filename))
            transformed_images.append(transformed_img)
            total_transforms += 1
        if _progress and (i + 1) % 1000 == 0:
            print(f"Processed {i + 1}/{len(images)} original images, created {total_transforms} transformed images")
    return transformed_images


def apply_all_transformations_batched(images):
    """`apply_all_transformations_batched_named` + the reference's save step (:159-162).  With `output_dir` set the files are
    written by the device writer (byte-identical to Pillow's `save`, tests/test_gpu_jpeg.py) while the images still come back;
    `IMGXF_SAVE=pillow` keeps Pillow's encoder."""
    if output_dir is not None and os.environ.get("IMGXF_SAVE", "device") == "device":
        return [img for _, img in _batched_to_files(images, output_dir, tee=True)]
    transformed_images = []
    for new_filename, img in apply_all_transformations_batched_named(images):
        if output_dir is not None:
            save_image(img, os.path.join(output_dir, new_filename))
        transformed_images.append(img)
    return transformed_images


def apply_all_transformations_batched_to_files(images, out_dir: str) -> List[str]:
    """The batched driver with the save step on the device and NO images copied back: returns the file names in output order
    (`_batched_to_files`)."""
    return [name for name, _ in _batched_to_files(images, out_dir, tee=False)]


def _batched_to_files(images, out_dir: str, tee: bool):
    """The batched driver with the save step (:159-162) on the device too: every group's result goes from the transform
    kernels straight into the JPEG writer (`jpeg.encode`), and only the files — a tenth of a byte per pixel for
    photographs instead of three — cross PCIe.  Same draws, names, order and FILES as `apply_all_transformations_batched`
    with `output_dir` set (Pillow's encoder): tests/test_gpu_jpeg.py.  Returns [(file name, image or None)] in output order:
    `tee` also copies the results back as PIL images (apply_all_transformations_batched with `output_dir` set)."""
    from . import jpeg
    os.makedirs(out_dir, exist_ok=True)
    # Groups are small (a handful of frames per transformation type and parameter) and a writer call costs ~0.4 ms of host
    # time whatever its size: results are collected per frame shape and encoded SINK_FRAMES at a time.
    held: dict = {}                                       # frame shape -> ([tensors], [names], frames)
    written: set = set()                                  # names the sink has put on disk

    def flush(shape) -> None:
        # (the files are written here, by this thread: a side thread pays a GIL hand-over per system call while this one
        # computes — 110 us per file against 19 us — and several threads queue on the directory's lock, 213 us per file)
        tensors, names, _ = held.pop(shape)
        big = tensors[0] if len(tensors) == 1 else torch.cat(tensors)
        for name, data in zip(names, jpeg.encode_views(big)):
            with open(os.path.join(out_dir, name), "wb") as f:
                f.write(data)
        written.update(names)

    def sink(out: torch.Tensor, names: List[str]) -> None:
        on_device = [n.lower().endswith((".jpg", ".jpeg")) for n in names]
        if all(on_device) and out.dim() == 4 and out.shape[-1] == 3:
            shape = tuple(out.shape[1:])
            slot = held.setdefault(shape, [[], [], 0])
            slot[0].append(out); slot[1].extend(names); slot[2] += out.shape[0]
            if slot[2] >= SINK_FRAMES or slot[2] * out[0].numel() >= SINK_BYTES:
                flush(shape)
        else:                                             # another format: Pillow writes it
            host = staging.download(out).numpy()
            for j, name in enumerate(names):
                Image.fromarray(host[j]).save(os.path.join(out_dir, name))
            written.update(names)

    named = apply_all_transformations_batched_named(images, _sink=sink, _tee=tee)
    sunk = set()
    if tee:                                               # the names the sink has written: everything else is saved below
        sunk = {n for _, names, _ in held.values() for n in names} | written
    for name, img in named:
        if isinstance(img, torch.Tensor):                 # apply_blur's radius-0 pass-through of a device frame
            sink(img[None], [name])
        elif img is not None and name not in sunk:        # per-image path (not RGB) or apply_blur's radius-0 pass-through
            save_image(img, os.path.join(out_dir, name))
    for shape in list(held):
        flush(shape)
    return named


SINK_FRAMES = 1024                                        # frames of one shape per writer call in the device-save drivers ...
SINK_BYTES = 2 << 30                                      # ... or this many bytes of them (4K frames: 86 per call), whichever comes first


def _size_of(img):
    """(width, height) of a PIL image or of an [H, W, 3] device frame (the device JPEG reader's output)."""
    return (int(img.shape[1]), int(img.shape[0])) if isinstance(img, torch.Tensor) else img.size


def _is_rgb(img) -> bool:
    return (img.dim() == 3 and img.shape[-1] == 3 and img.dtype == torch.uint8) if isinstance(img, torch.Tensor) else img.mode == 'RGB'


def _collect(item, results) -> int:
    """Wait for one queued copy back and build its PIL images; returns the bytes it held."""
    dl, entries, *flag = item                             # (download, entries[, frames are RGBX for staging.image_from_rgbx])
    rgbx = bool(flag and flag[0])
    host = dl.numpy()
    for j, (_, i, k) in enumerate(entries):
        results[i][k] = staging.image_from_rgbx(host[j]) if rgbx else Image.fromarray(host[j])
    return host.nbytes


def apply_all_transformations_batched_named(images, _sink=None, _tee=False):
    """`apply_all_transformations` with the work grouped for the GPU, returning
    [(file name, image)] in the reference's output order and saving nothing: same draws (`random` per
    transform type per image, `np.random` for the noise, in the reference's order), same file
    names, same outputs in the same order — but every image is uploaded once, and all images
    of one size that drew the same (type, value) go through ONE batched launch.  Images that
    are not 8-bit RGB take the per-image path.  images: [(PIL image, path)] — or [(frame, path)] with [H, W, 3] uint8
    DEVICE tensors as the device JPEG reader returns them (`jpeg_decode.decode`): those are never copied to the host.  `_sink(out, names)`, when given,
    consumes a group's result ON THE DEVICE ([B, H, W, 3] tensor + its file names) instead of it being copied back:
    those entries come back as (file name, None)."""
    dev = _device()
    plans, noise = [], {}
    draws = []                                          # (i, k, (h, w, 3), scale): np.random.normal calls in the per-image loop's order
    for i, (img, path) in enumerate(images):
        name = os.path.splitext(os.path.basename(path))[0]
        plan = plan_transformations(name)
        plans.append(plan)
        for k, (transform_type, args, _) in enumerate(plan):
            if transform_type == 'gaussian_noise':      # same np.random stream as the per-image loop
                w, h = _size_of(img)
                if NOISE_RNG == "device":               # opt-in: one seed per image instead of h * w * 3 normals
                    noise[(i, k)] = _noise_seed()
                else:
                    draws.append((i, k, (h, w, 3), args[0] * 255))
    pending_noise = None
    if draws:
        # nothing else touches np.random between these calls (the grid values come from `random`): one pass over the stream
        # on the device serves them all, or the host makes them one by one.  The MT19937 block kernel (one workgroup) starts
        # NOW on a side stream; the numbers are collected when the first noise group comes up, after the other transformation
        # types have been queued (tensor_fns order puts 'gaussian_noise' wherever the reference's dict has it).
        reqs = [(h * w * c, scale) for _, _, (h, w, c), scale in draws]
        if NOISE_RNG != "numpy-host" and sum(n for n, _ in reqs) >= NOISE_DEVICE_MIN:
            from . import numpy_stream
            pending_noise = numpy_stream.PendingDraw(reqs, dev)
        else:
            for (i, k, shape, scale) in draws:
                noise[(i, k)] = np.random.normal(0, scale, shape).astype(np.float32)

    def collect_noise():
        nonlocal pending_noise
        if pending_noise is None:
            return
        p, pending_noise = pending_noise, None
        state = p.state
        try:
            got = p.result()
        except ValueError:                              # (a margin of 12 standard deviations was too short: the host draws)
            np.random.set_state(state)
            got = None
        for n_, ((i, k, shape, scale)) in enumerate(draws):
            z = got[n_] if got is not None else None
            noise[(i, k)] = z.view(shape) if z is not None else np.random.normal(0, scale, shape).astype(np.float32)

    results = [[None] * len(p) for p in plans]
    by_size = {}
    for i, (img, _) in enumerate(images):
        if _is_rgb(img):
            by_size.setdefault(_size_of(img), []).append(i)
        else:                                           # rare: keep the reference's behaviour exactly
            for k, (transform_type, args, _) in enumerate(plans[i]):
                if transform_type == 'gaussian_noise':
                    raise NotImplementedError("batched driver expects RGB images (load_data converts them)")
                fn = apply_translation if transform_type == 'translation' else _DISPATCH[transform_type]
                results[i][k] = fn(img, *args)

    tensor_fns = {
        'scale': _scale_t,
        'rotation': _rotation_t,
        'lighten_darken': lambda t, b: ops.brightness(t, 1.0 + b),
        'contrast': lambda t, a: ops.scale_abs(t, a, 0.0),
        'shear': _shear_t,
        'translation': _translation_t,
    }
    pending, queued = [], 0                             # (Download, entries): results still on their way back
    for size, members in by_size.items():
        if all(isinstance(images[i][0], torch.Tensor) for i in members):            # already on the device (JPEG reader)
            frames = torch.stack([images[i][0] for i in members])
        else:
            frames = staging.upload([np.asarray(images[i][0].cpu() if isinstance(images[i][0], torch.Tensor) else images[i][0])
                                     for i in members], dev)                        # one pinned block, async H2D
        groups = {}
        for row, i in enumerate(members):
            for k, (transform_type, args, _) in enumerate(plans[i]):
                groups.setdefault((transform_type, args), []).append((row, i, k))
        # (noise groups last: their numbers come from the side stream's generator, which runs meanwhile)
        ordered = sorted(groups.items(), key=lambda g: g[0][0] == 'gaussian_noise')
        for (transform_type, args), entries in ordered:
            rows = torch.tensor([e[0] for e in entries], device=dev)
            batch = frames.index_select(0, rows)
            if transform_type == 'blur':
                ksize = _blur_ksize(args[0])
                if ksize == 0:
                    for _, i, k in entries:
                        results[i][k] = images[i][0]    # the input object itself (:245-246)
                    continue
                out = ops.gaussian_blur(batch, ksize, args[0], fixed_point=BLUR_FIXED_POINT)
            elif transform_type == 'gaussian_noise' and NOISE_RNG == "device":
                out = torch.empty_like(batch)           # every image has its own seed (as the per-image call draws it)
                for j, (_, i, k) in enumerate(entries):
                    out[j] = ops.add_noise_device(batch[j], args[0] * 255, noise[(i, k)])
            elif transform_type == 'gaussian_noise':
                collect_noise()
                zs = [noise[(i, k)] for _, i, k in entries]
                z = torch.stack(zs) if isinstance(zs[0], torch.Tensor) else staging.upload(zs, dev)
                out = ops.add_noise(batch, z)
            else:
                out = tensor_fns[transform_type](batch, *args)
            if _sink is not None:
                _sink(out, [plans[i][k][2] for _, i, k in entries])
                if not _tee:
                    continue
            # queue the copy back and keep launching: the host waits per result only when it builds the images.
            # The window of copies in flight is bounded (staging.PENDING_BUDGET bytes of pinned memory): beyond
            # it the oldest results are turned into images before the next group is queued
            # RGB frames go back as RGBX and become PIL images that share the pinned block (staging.image_from_rgbx)
            rgbx = out.dim() == 4 and out.shape[-1] == 3 and out.dtype == torch.uint8 and \
                staging.zero_copy_reserve(out.shape[0] * out.shape[1] * out.shape[2] * 4)
            pending.append((staging.download(ops.permute_channels(out, (0, 1, 2, 2)) if rgbx else out), entries, rgbx))
            queued += out.numel()
            while queued > staging.PENDING_BUDGET and len(pending) > 1:
                queued -= _collect(pending.pop(0), results)
    while pending:
        _collect(pending.pop(0), results)

    return [(new_filename, results[i][k]) for i, plan in enumerate(plans) for k, (_, _, new_filename) in enumerate(plan)]
