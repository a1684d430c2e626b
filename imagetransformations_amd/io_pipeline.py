"""The on-disk steps either side of the hot path (/root/reference/transformation.py:73-89 load,
:159-162 save) as a streaming pipeline around the batched driver: JPEG decode and encode stay on
the CPU with Pillow — the same codec the reference uses, so the pixels entering and leaving are the
reference's — but run in worker threads (Pillow releases the GIL inside the codecs) that overlap
with the GPU work of the neighbouring chunks (`encoder="device"` moves the encode of RGB → *.jpg outputs
to the GPU writer, `jpeg.encode`, whose files are byte-identical to Pillow's):

    decode chunk k+1  |  transform chunk k on the GPU  |  encode + write chunk k-1

Chunks are consecutive slices of the file list in the reference's order, and the batched driver
draws per image in list order, so the `random` / `np.random` streams — hence file names and pixels
— are those of the reference's one-image-at-a-time loop."""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Iterable, List, Sequence, Tuple

from PIL import Image


def list_images(data_path: str, suffix: str = ".jpeg") -> List[str]:
    """The file order of the reference's load_data (:74-78): os.walk order, *.jpeg, case-insensitive."""
    paths = []
    for root, _, files in os.walk(data_path):
        paths.extend(os.path.join(root, f) for f in files if f.lower().endswith(suffix))
    return paths


def _decode(path: str):
    try:
        return Image.open(path).convert("RGB"), path          # :83
    except Exception as e:                                     # the reference guards only the loading (:85-86)
        print(f"Failed to load image {path}: {e}")
        return None


def _read(path: str):
    try:
        with open(path, "rb") as f:
            return f.read(), path
    except Exception as e:
        print(f"Failed to load image {path}: {e}")
        return None


DECODE_STATS = {"device": 0, "pillow": 0}      # files decoded by the device reader / handed to Pillow by decoder="device"


def _decode_on_device(read):
    """[(bytes, path)] -> [(frame, path)]: one `jpeg_decode.decode` over the chunk; a file outside the device reader's
    class (progressive, CMYK, ...: `UnsupportedJpeg`) or one it finds damaged is decoded by Pillow, as :83 does, and
    uploaded — counted in DECODE_STATS so that the share is visible."""
    import io
    import numpy as np
    import torch
    from . import jpeg_decode
    from ._ffi import ImgxfError
    items = [r for r in read if r is not None]
    try:
        frames = jpeg_decode.decode([d for d, _ in items])
        DECODE_STATS["device"] += len(items)
        return [(t, p) for t, (_, p) in zip(frames, items)]
    except (jpeg_decode.UnsupportedJpeg, ImgxfError):
        out = []
        for d, p in items:
            try:
                out.append((jpeg_decode.decode([d])[0], p))
                DECODE_STATS["device"] += 1
            except (jpeg_decode.UnsupportedJpeg, ImgxfError):
                try:
                    img = Image.open(io.BytesIO(d)).convert("RGB")
                    out.append((torch.from_numpy(np.asarray(img)).cuda(), p))
                    DECODE_STATS["pillow"] += 1
                except Exception as e:
                    print(f"Failed to load image {p}: {e}")
        return out


def _chunks(seq: Sequence, n: int) -> Iterable[Sequence]:
    for i in range(0, len(seq), n):
        yield seq[i:i + n]


def run_directory(data_path: str, out_dir: str, chunk_images: int = 256, workers: int = 8,
                  transform: Callable[[List[Tuple[Image.Image, str]]], List[Tuple[str, Image.Image]]] | None = None,
                  encoder: str = "device", decoder: str | None = None) -> int:
    """load_data + apply_all_transformations + save over a directory, streamed.  `transform` maps a
    chunk [(image, path)] to [(file name, image)] in output order; default: the batched
    eight-transformation driver.  Returns the number of files written.  Both on-disk steps default to the device since
    round 3 (files byte-identical to Pillow's, pixels identical to Pillow's; "pillow" selects the host libraries the
    reference uses; with a caller's own `transform` the decoder defaults to "pillow", since it receives PIL images).  `decoder="device"`: worker threads only READ the
    files; the chunk is decoded by the GPU reader (`jpeg_decode.decode`, pixels identical to Pillow's) and the frames
    go to the batched driver without ever visiting the host."""
    if decoder is None:                                         # a caller's own `transform` is written against PIL images
        decoder = "device" if transform is None else "pillow"
    device_driver = transform is None and encoder == "device"
    if transform is None:
        from .transformation import apply_all_transformations_batched_named as transform
        from .transformation import apply_all_transformations_batched_to_files as to_files
    if encoder not in ("pillow", "device"):
        raise ValueError("encoder must be 'pillow' or 'device'")
    if decoder not in ("pillow", "device"):
        raise ValueError("decoder must be 'pillow' or 'device'")
    load = _read if decoder == "device" else _decode
    os.makedirs(out_dir, exist_ok=True)
    chunk_paths = list(_chunks(list_images(data_path), chunk_images))
    written = 0
    with ThreadPoolExecutor(max_workers=workers) as pool:
        decoding = pool.map(load, chunk_paths[0]) if chunk_paths else None
        saving = []
        for k in range(len(chunk_paths)):
            chunk = [d for d in decoding if d is not None]
            if k + 1 < len(chunk_paths):                        # the next chunk decodes (is read) while this one is on the GPU
                decoding = pool.map(load, chunk_paths[k + 1])
            if decoder == "device":
                chunk = _decode_on_device(chunk)
            if device_driver:                                   # transforms + JPEG writer, nothing but the files comes back
                for fut in saving:
                    fut.result()
                saving = []
                written += len(to_files(chunk, out_dir)) if chunk else 0
                continue
            named = transform(chunk) if chunk else []
            if decoder == "device":                             # (apply_blur's radius-0 pass-through hands a device frame back)
                named = [(nm, Image.fromarray(im.cpu().numpy()) if hasattr(im, "cpu") else im) for nm, im in named]
            for fut in saving:                                  # chunk k-1 has been encoding meanwhile
                fut.result()
            written += len(named)
            saving = []
            if encoder == "device":
                named = _save_on_device(named, out_dir, pool, saving)
            saving += [pool.submit(_save, img, os.path.join(out_dir, name)) for name, img in named]
        for fut in saving:
            fut.result()
    return written


def _save(img: Image.Image, path: str) -> None:
    img.save(path)                                              # Image.save defaults, as :162


def _write(data: bytes, path: str) -> None:
    with open(path, "wb") as f:
        f.write(data)


def _save_on_device(named, out_dir: str, pool, saving: list):
    """Encode the RGB images bound for *.jpg / *.jpeg files with the GPU writer, one batch per image size (the files are
    Pillow's, byte for byte: tests/test_gpu_jpeg.py); queue the writes on `pool`; return what is left for Pillow
    (other modes / formats, images carrying a comment Pillow would embed)."""
    import numpy as np
    import torch
    from . import jpeg
    groups, rest = {}, []
    for name, img in named:
        if img.mode == "RGB" and name.lower().endswith((".jpg", ".jpeg")) and "comment" not in img.info and min(img.size) > 0:
            groups.setdefault(img.size, []).append((name, img))
        else:
            rest.append((name, img))
    for items in groups.values():
        batch = torch.from_numpy(np.stack([np.asarray(img) for _, img in items])).cuda(non_blocking=True)
        for (name, _), data in zip(items, jpeg.encode_views(batch)):
            saving.append(pool.submit(_write, data, os.path.join(out_dir, name)))
    return rest


SEVERITY_INDICES = [0, 1001, 2002, 3003, 4004, 10000, 10001, 12002, 13003, 14004, 15005, 20000, 22002, 23003,
                    24004, 25005, 30000, 40000]          # transformation.py:28 (zip with five labels uses the first five)
SEVERITY_LABELS = [1, 2, 3, 4, 5]


def load_data_npy(data_path: str, store_dir: str) -> int:
    """The CIFAR-10-C extraction step (transformation.py:20-71): every *.npy under `data_path`
    with shape (50000, 32, 32, 3) contributes the images at the first five severity indices,
    written as `{corruption}_severity{level}_idx{index}.png`.  The arrays are memory-mapped
    (153 MB each; five 3 KiB slices are read), the bytes written are the reference's."""
    import numpy as np
    os.makedirs(store_dir, exist_ok=True)
    npy_files = []
    for root, _, files in os.walk(data_path):
        npy_files.extend(os.path.join(root, f) for f in files if f.lower().endswith('.npy'))
    extracted_count = 0
    for file_path in npy_files:
        try:
            corruption_data = np.load(file_path, mmap_mode="r")
            corruption_name = os.path.splitext(os.path.basename(file_path))[0]
            if corruption_data.shape != (50000, 32, 32, 3):
                print(f"  Warning: Unexpected shape for {corruption_name}.npy: {corruption_data.shape}. Skipping.")
                continue
            for s_idx, severity_level in zip(SEVERITY_INDICES, SEVERITY_LABELS):
                img_array = np.array(corruption_data[s_idx])
                if img_array.dtype != np.uint8:
                    img_array = img_array.astype(np.uint8)
                Image.fromarray(img_array).save(
                    os.path.join(store_dir, f"{corruption_name}_severity{severity_level}_idx{s_idx}.png"))
                extracted_count += 1
        except Exception as e:
            print(f"Failed to process file {file_path}: {e}")
    return extracted_count
