"""GPU: the device JPEG reader (csrc/jpeg_decode.hip + imagetransformations_amd/jpeg_decode.py) — the decode half of
the reference's load step `Image.open(path).convert("RGB")` (/root/reference/transformation.py:83) — against Pillow /
libjpeg-turbo itself and the NumPy oracle: bit-identical pixels on the 30 files the reference wrote
(tests/golden/reference_outputs/) and on seeded images of every size class, sampling, quality and table kind."""
import glob
import io
import os

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import synth
from oracle import jpeg_decode_oracle as JD
from test_jpeg_decode_oracle import photo_like, pillow_rgb

pytestmark = pytest.mark.gpu
REF = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "reference_outputs", "*.JPEG")))


def jpeg_bytes(img, **kw):
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, "JPEG", **kw)
    return buf.getvalue()


def test_reference_written_files_in_one_batch(device):
    from imagetransformations_amd import jpeg_decode
    files = [open(p, "rb").read() for p in REF]
    assert len(files) == 30
    frames = jpeg_decode.decode(files, device)
    for p, data, t in zip(REF, files, frames):
        want = pillow_rgb(data)
        assert t.device.type == "cuda" and tuple(t.shape) == want.shape
        assert np.array_equal(t.cpu().numpy(), want), os.path.basename(p)
    assert np.array_equal(frames[3].cpu().numpy(), JD.decode(files[3]))


@pytest.mark.parametrize("subsampling", [0, 1, 2])
def test_sizes_samplings_qualities(device, subsampling):
    from imagetransformations_amd import jpeg_decode
    files, wants = [], []
    for (h, w) in [(1, 1), (7, 5), (8, 8), (16, 16), (17, 33), (31, 15), (48, 64), (100, 75), (375, 500), (123, 457),
                   (9, 2), (40, 3), (174, 4), (21, 6), (2, 40)]:                # narrow files: libjpeg replicates chroma when its width <= 2
        for seed, quality in ((1, 75), (2, 30), (3, 95)):
            for img in (synth(seed * 7 + h, h, w), photo_like(seed, h, w)):
                files.append(jpeg_bytes(img, quality=quality, subsampling=subsampling))
                wants.append(pillow_rgb(files[-1]))
    frames = jpeg_decode.decode(files, device)
    for k, (t, want) in enumerate(zip(frames, wants)):
        assert np.array_equal(t.cpu().numpy(), want), (k, want.shape)


def test_grayscale_optimised_tables_restart_intervals_and_groups(device):
    from imagetransformations_amd import jpeg_decode
    img = photo_like(5, 161, 283)
    files = [jpeg_bytes(img, optimize=True), jpeg_bytes(img, quality=10), jpeg_bytes(img, quality=100, subsampling=0),
             jpeg_bytes(img, restart_marker_blocks=3), jpeg_bytes(img, restart_marker_rows=1, subsampling=2),
             jpeg_bytes(img, restart_marker_blocks=1, subsampling=1)]
    buf = io.BytesIO(); Image.fromarray(img).convert("L").save(buf, "JPEG", quality=80); files.append(buf.getvalue())
    buf = io.BytesIO(); Image.fromarray(img).convert("L").save(buf, "JPEG", quality=50, optimize=True, restart_marker_rows=2); files.append(buf.getvalue())
    frames = jpeg_decode.decode(files, device)
    for k, (t, f) in enumerate(zip(frames, files)):
        assert np.array_equal(t.cpu().numpy(), pillow_rgb(f)), k
    # equal-sized files share one allocation: the frames are views of a [N, H, W, 3] batch
    assert frames[0].untyped_storage().data_ptr() == frames[1].untyped_storage().data_ptr()
    groups = jpeg_decode.decode_batches(files, device)
    (batch, members), = groups.values()
    assert tuple(batch.shape) == (len(files), 161, 283, 3) and members == list(range(len(files)))


def test_full_hd_and_4k_frames(device):
    from imagetransformations_amd import jpeg_decode
    files = [jpeg_bytes(photo_like(9, 1080, 1920)), jpeg_bytes(photo_like(10, 2160, 3840), restart_marker_rows=1),
             jpeg_bytes(synth(11, 540, 960), quality=90)]
    frames = jpeg_decode.decode(files, device)
    for t, f in zip(frames, files):
        assert np.array_equal(t.cpu().numpy(), pillow_rgb(f))


def test_unsupported_and_damaged_files(device):
    from imagetransformations_amd import _ffi, jpeg_decode
    img = photo_like(1, 64, 64)
    with pytest.raises(jpeg_decode.UnsupportedJpeg):
        jpeg_decode.decode([jpeg_bytes(img, progressive=True)], device)
    buf = io.BytesIO(); Image.fromarray(img).convert("CMYK").save(buf, "JPEG")
    with pytest.raises(jpeg_decode.UnsupportedJpeg):
        jpeg_decode.decode([buf.getvalue()], device)
    with pytest.raises(jpeg_decode.UnsupportedJpeg):
        jpeg_decode.decode([b"not a jpeg at all"], device)
    good = jpeg_bytes(img, optimize=True)
    info = jpeg_decode.parse(good)
    cut = good[:info["ecs"][0] + 40] + b"\xff\xd9"                      # the scan stops after 40 bytes
    with pytest.raises(_ffi.ImgxfError):
        jpeg_decode.decode([good, cut], device)
    assert jpeg_decode.decode([], device) == []
    with pytest.raises(_ffi.ImgxfError):
        jpeg_decode.decode([good], "cpu")                                # there is no CPU fallback


def test_decode_then_transform_stays_on_the_device(device):
    """load -> transform without a host round trip: the reader's frames feed the ops directly."""
    from imagetransformations_amd import jpeg_decode, ops
    files = [open(p, "rb").read() for p in REF[:4]]
    frames = jpeg_decode.decode(files, device)
    for t, f in zip(frames, files):
        from oracle import imgxf_oracle as O
        a = pillow_rgb(f)
        assert np.array_equal(ops.rotate(t, -22.5, ops.NEAREST, (0, 0, 0)).cpu().numpy(), O.apply_rotation(a, 22.5))


def test_parallel_and_serial_entropy_decoders_agree_with_pillow(device, monkeypatch):
    """Long restart segments (files without restart markers are ONE segment) are decoded by a workgroup per image that
    synchronises 256 / 1024 speculative subsequences (jpeg_huff_par_kernel), several mid-size segments by a wave each; short
    segments keep a lane each.  Both must give
    Pillow's pixels: every sampling, grayscale, optimised tables, scans of one to several 256-subsequence chunks, restart
    intervals that leave long segments, mixed batches — and the same answer with the parallel decoder switched off."""
    from imagetransformations_amd import _ffi, jpeg_decode
    files = []
    for i, (h, w, kw) in enumerate([(375, 500, dict(subsampling=2)), (375, 500, dict(subsampling=0, quality=95)), (480, 640, dict(subsampling=1)),
                                    (1080, 1920, dict(subsampling=2, quality=90)), (1080, 1920, dict(subsampling=2, restart_marker_rows=17)),
                                    (333, 517, dict(optimize=True)), (768, 1024, dict(quality=30)), (64, 64, {}), (1200, 1600, dict(quality=98, subsampling=0)),
                                    (600, 800, dict(restart_marker_blocks=700)),
                                    (1080, 1920, dict(restart_marker_rows=1, quality=90)),          # 68 segments of a few KB: a wave per segment
                                    (720, 1280, dict(restart_marker_rows=2, subsampling=0)), (400, 2000, dict(restart_marker_blocks=40, quality=95))]):
        files.append(jpeg_bytes(photo_like(60 + i, h, w), **kw))
    buf = io.BytesIO(); Image.fromarray(photo_like(77, 900, 1200)).convert("L").save(buf, "JPEG", quality=85); files.append(buf.getvalue())
    want = [np.asarray(Image.open(io.BytesIO(f)).convert("RGB")) for f in files]
    got = [t.cpu().numpy() for t in jpeg_decode.decode(files, device)]
    for i, (g, wnt) in enumerate(zip(got, want)):
        assert np.array_equal(g, wnt), i
    monkeypatch.setenv("IMGXF_JPEG_SERIAL_HUFFMAN", "1")
    got1 = [t.cpu().numpy() for t in jpeg_decode.decode(files, device)]
    monkeypatch.delenv("IMGXF_JPEG_SERIAL_HUFFMAN")
    for i, (g, g1) in enumerate(zip(got, got1)):
        assert np.array_equal(g, g1), i
    # damaged long scans are reported, not decoded into garbage silently: cut in the middle, and a corrupted stretch
    big = files[3]
    info = jpeg_decode.parse(big)
    s0, s1 = info["ecs"]
    cut = big[:s0 + (s1 - s0) // 2] + b"\xff\xd9"
    with pytest.raises(_ffi.ImgxfError):
        jpeg_decode.decode([files[0], cut], device)


def test_corrupted_scans_are_reported_or_decoded_never_fatal(device):
    """Byte flips, random stretches and truncations inside the entropy-coded data, for files of both decoder classes (one long
    segment / a restart marker per MCU row): every call returns frames or raises; afterwards good files still decode."""
    from imagetransformations_amd import _ffi, jpeg_decode
    rng = np.random.default_rng(3)
    img = photo_like(90, 400, 560)
    seeds = [jpeg_bytes(img, quality=85), jpeg_bytes(img, quality=85, restart_marker_rows=1), jpeg_bytes(img, quality=60, subsampling=0)]
    reported = 0
    for c in range(24):
        batch = []
        for f in seeds:
            s0, s1 = jpeg_decode.parse(f)["ecs"]
            g = bytearray(f)
            if c % 3 == 0:
                for _ in range(int(rng.integers(1, 30))):
                    g[int(rng.integers(s0, s1))] = int(rng.integers(0, 256))
            elif c % 3 == 1:
                g = g[:int(rng.integers(s0, s1))] + b"\xff\xd9"
            else:
                a = int(rng.integers(s0, s1 - 64)); g[a:a + 64] = bytes(rng.integers(0, 256, 64, dtype=np.uint8))
            batch.append(bytes(g))
        try:
            jpeg_decode.decode(batch, device)
        except (_ffi.ImgxfError, jpeg_decode.UnsupportedJpeg):
            reported += 1
    assert reported >= 12
    for t, f in zip(jpeg_decode.decode(seeds, device), seeds):
        assert np.array_equal(t.cpu().numpy(), np.asarray(Image.open(io.BytesIO(f)).convert("RGB")))
