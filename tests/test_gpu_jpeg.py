"""Device JPEG writer (`imagetransformations_amd.jpeg`, `imgxf_jpeg_encode_u8`) against Pillow's own encoder — the library
behind the reference driver's `transformed.save(path)` (transformation.py:161-162) — the CPU restatement and the
committed fixtures: whole files, byte for byte."""
import hashlib, importlib.util, io, json, os
import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_jpeg_golden", os.path.join(HERE, "golden", "make_jpeg_golden.py"))
G = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(G)
GOLDEN = json.load(open(os.path.join(HERE, "golden", "jpeg_q75.json")))


def pil_bytes(a, **kw):
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, "JPEG", **kw)
    return buf.getvalue()


def gpu_bytes(a, quality=75, **kw):
    from imagetransformations_amd import jpeg
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return jpeg.encode(t if t.dim() == 4 else t[None], quality, **kw)


def first_diff(a, b):
    n = min(len(a), len(b))
    return next((i for i in range(n) if a[i] != b[i]), n)


@pytest.mark.parametrize("case", GOLDEN["cases"], ids=lambda c: f"{c['h']}x{c['w']}-{c['kind']}-q{c['quality']}")
def test_fixture(case):
    img = G.image(case["h"], case["w"], case["kind"], case["seed"])
    out = gpu_bytes(img, 75 if case["quality"] is None else case["quality"])[0]
    assert len(out) == case["bytes"]
    assert hashlib.sha256(out).hexdigest() == case["sha256"]


@pytest.mark.parametrize("shape", [(1, 1), (2, 3), (7, 9), (8, 8), (15, 17), (16, 16), (16, 256), (16, 257), (17, 255), (31, 300),
                                   (33, 511), (64, 48), (100, 75), (375, 500), (32, 32), (240, 1000), (1080, 1920)])
def test_equals_pillow_and_oracle(shape):
    from oracle import jpeg_oracle as J
    rng = np.random.default_rng(shape[0] * 7919 + shape[1])
    for kind in range(3):
        if kind == 0:
            a = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
        elif kind == 1:
            yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
            a = np.stack([(xx * 3 + yy) % 256, (xx + yy * 2) % 256, (xx * yy) % 256], -1).astype(np.uint8)
        else:
            a = np.full(shape + (3,), rng.integers(0, 256), np.uint8)
            a[shape[0] // 2:, :, 1] = 255
        want, got = pil_bytes(a), gpu_bytes(a)[0]
        assert len(got) == len(want) and got == want, f"kind {kind}: first difference at byte {first_diff(got, want)} of {len(want)}"
        if shape[0] * shape[1] <= 200000:
            assert J.encode(a) == got


@pytest.mark.parametrize("quality", [1, 10, 50, 90, 95, 100])
def test_qualities(quality):
    rng = np.random.default_rng(quality)
    a = rng.integers(0, 256, (75, 130, 3), dtype=np.uint8)
    assert gpu_bytes(a, quality)[0] == pil_bytes(a, quality=quality)
    b = (rng.integers(0, 256, (75, 130, 3)) // 64 * 60 + 10).astype(np.uint8)
    assert gpu_bytes(b, quality)[0] == pil_bytes(b, quality=quality)


def test_batch_frames_are_independent_files():
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (9, 50, 70, 3), dtype=np.uint8)
    a[3] = 0
    a[4] = 255
    a[5, :, :, :] = np.arange(70, dtype=np.uint8)[None, :, None] * 3
    out = gpu_bytes(a)
    assert len(out) == 9
    for i in range(9):
        assert out[i] == pil_bytes(a[i]), i


def test_strided_and_unaligned_views():
    from imagetransformations_amd import jpeg
    rng = np.random.default_rng(6)
    big = torch.from_numpy(rng.integers(0, 256, (3, 300, 600, 3), dtype=np.uint8)).cuda()
    view = big[:, 5:277, 3:515, :]                      # row stride 1800, base offset 9 bytes: not 16-byte aligned
    out = jpeg.encode(view)
    for i in range(3):
        assert out[i] == pil_bytes(view[i].cpu().numpy())
    aligned = torch.from_numpy(rng.integers(0, 256, (2, 64, 512, 3), dtype=np.uint8)).cuda()
    out = jpeg.encode(aligned)
    for i in range(2):
        assert out[i] == pil_bytes(aligned[i].cpu().numpy())


def test_4k_frame():
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:2160, 0:3840]
    a = np.stack([(xx // 7 + yy // 3) % 256, (xx // 2 + yy // 5) % 256, (xx // 11 * 3 + yy // 13) % 256], -1).astype(np.uint8)
    a[500:900, 700:1900] = rng.integers(0, 256, (400, 1200, 3), dtype=np.uint8)
    got = gpu_bytes(a)[0]
    want = pil_bytes(a)
    assert got == want, first_diff(got, want)
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(got))), np.asarray(Image.open(io.BytesIO(want))))


def test_capacity_is_enforced_loudly_and_default_grows():
    from imagetransformations_amd import jpeg, _ffi
    rng = np.random.default_rng(8)
    a = torch.from_numpy(rng.integers(0, 256, (2, 64, 64, 3), dtype=np.uint8)).cuda()
    with pytest.raises(_ffi.ImgxfError):
        jpeg.encode(a, 100, capacity=2048)
    files, sizes = jpeg.encode_device(a, 100, capacity=2048)
    assert sizes.cpu().tolist() == [0xFFFFFFFF] * 2
    out = jpeg.encode(a, 100)                                  # q=100 noise: > 2 bytes per pixel, the default retries larger
    assert out[0] == pil_bytes(a[0].cpu().numpy(), quality=100)


def test_argument_errors():
    from imagetransformations_amd import jpeg
    with pytest.raises(ValueError):
        jpeg.encode(torch.zeros((1, 8, 8, 4), dtype=torch.uint8, device="cuda"))
    with pytest.raises(ValueError):
        jpeg.encode(torch.zeros((1, 8, 8, 3), dtype=torch.float32, device="cuda"))
    assert jpeg.encode(torch.zeros((0, 8, 8, 3), dtype=torch.uint8, device="cuda")) == []


def test_run_directory_device_encoder_writes_pillows_files(tmp_path):
    from imagetransformations_amd import io_pipeline
    rng = np.random.default_rng(11)
    src = tmp_path / "in"
    src.mkdir()
    for i, (h, w) in enumerate([(40, 60), (40, 60), (33, 47), (64, 64), (40, 60)]):
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(src / f"img{i}.jpeg", quality=95)

    def transform(chunk):
        out = []
        for img, path in chunk:
            stem = os.path.splitext(os.path.basename(path))[0]
            out.append((f"{stem}_brightness_0.5_corrupted.jpg", img))
            out.append((f"{stem}_mask.png", img.convert("L")))
        return out

    counts = {}
    for enc in ("pillow", "device"):
        counts[enc] = io_pipeline.run_directory(str(src), str(tmp_path / enc), chunk_images=2, workers=2, transform=transform, encoder=enc)
    assert counts["pillow"] == counts["device"] == 10
    names = sorted(os.listdir(tmp_path / "pillow"))
    assert names == sorted(os.listdir(tmp_path / "device")) and len(names) == 10
    for n in names:
        assert (tmp_path / "pillow" / n).read_bytes() == (tmp_path / "device" / n).read_bytes(), n


def test_facade_save_image_switch(tmp_path, monkeypatch):
    from imagetransformations_amd import transformation as T
    rng = np.random.default_rng(12)
    img = Image.fromarray(rng.integers(0, 256, (45, 70, 3), dtype=np.uint8))
    gray = img.convert("L")
    monkeypatch.setattr(T, "JPEG_ON_DEVICE", True)
    T.save_image(img, str(tmp_path / "a_rotation_10.0_corrupted.jpg"))
    T.save_image(gray, str(tmp_path / "g.jpg"))                      # not RGB: Pillow writes it
    T.save_image(img, str(tmp_path / "a.png"))
    assert (tmp_path / "a_rotation_10.0_corrupted.jpg").read_bytes() == pil_bytes(np.asarray(img))
    assert (tmp_path / "g.jpg").read_bytes() == pil_bytes(np.asarray(gray))
    assert np.array_equal(np.asarray(Image.open(tmp_path / "a.png")), np.asarray(img))


def test_c_abi_argument_checks():
    import ctypes
    from imagetransformations_amd import _ffi as F, jpeg
    t = torch.zeros((1, 16, 16, 3), dtype=torch.uint8, device="cuda")
    t4 = torch.zeros((1, 16, 16, 4), dtype=torch.uint8, device="cuda")
    out = torch.zeros((4096,), dtype=torch.uint8, device="cuda")
    sizes = torch.zeros((1,), dtype=torch.int32, device="cuda")
    need = ctypes.c_size_t()
    F.call("imgxf_jpeg_workspace_bytes", 1, 16, 16, 4096, ctypes.byref(need))
    ws = torch.zeros((need.value,), dtype=torch.uint8, device="cuda")
    hdr = jpeg.header(16, 16)
    tabs = jpeg.tables(75)

    def run(view, tables=tabs, header=hdr, hlen=len(hdr), cap=4096, wsb=need.value):
        return F.lib.imgxf_jpeg_encode_u8(F.vp(F.view_of(view)), ctypes.addressof(tables), header, hlen, out.data_ptr(), cap,
                                          sizes.data_ptr(), ws.data_ptr(), wsb, None)

    assert run(t) == F.OK
    torch.cuda.synchronize()
    assert out[:sizes.item()].cpu().numpy().tobytes() == pil_bytes(np.zeros((16, 16, 3), np.uint8))
    assert run(t4) == F.ERR_UNSUPPORTED
    assert run(t, hlen=2000) == F.ERR_ARG
    assert run(t, cap=100) == F.ERR_ARG
    assert run(t, wsb=need.value - 1) == F.ERR_WORKSPACE
    bad = F.JpegTables.from_buffer_copy(tabs)
    bad.quant[1][5] = 0
    assert run(t, tables=bad) == F.ERR_ARG
    with pytest.raises(ValueError):
        F.call("imgxf_jpeg_workspace_bytes", 1, 0, 16, 4096, ctypes.byref(need))


def test_batched_driver_with_the_save_step_on_the_device(tmp_path, monkeypatch):
    """apply_all_transformations_batched_to_files == apply_all_transformations_batched + Pillow's save: same names, same
    files, for the same `random` / `np.random` seeds (mixed sizes; seeds that draw a radius-0 blur included)."""
    import random
    from imagetransformations_amd import transformation as T
    rng = np.random.default_rng(21)
    imgs = [(Image.fromarray(rng.integers(0, 256, hw + (3,), dtype=np.uint8)), f"/data/n0{i}/img_{i}.JPEG")
            for i, hw in enumerate([(32, 32), (48, 64), (32, 32), (37, 61), (48, 64)])]
    for seed in (0, 1, 2, 3):
        ref_dir, dev_dir = tmp_path / f"pillow{seed}", tmp_path / f"device{seed}"
        ref_dir.mkdir()
        monkeypatch.setattr(T, "output_dir", str(ref_dir))
        random.seed(seed); np.random.seed(seed)
        T.apply_all_transformations_batched(imgs)
        monkeypatch.setattr(T, "output_dir", None)
        random.seed(seed); np.random.seed(seed)
        names = T.apply_all_transformations_batched_to_files(imgs, str(dev_dir))
        assert len(names) == 8 * len(imgs)
        assert sorted(os.listdir(ref_dir)) == sorted(os.listdir(dev_dir)) == sorted(set(names))
        for n in set(names):
            assert (ref_dir / n).read_bytes() == (dev_dir / n).read_bytes(), (seed, n)


def test_run_directory_default_driver_device_encoder(tmp_path):
    """The whole on-disk loop (load_data → eight transforms per image → save) with the save step on the device writes
    the files the Pillow-encoder pipeline writes."""
    import random
    from imagetransformations_amd import io_pipeline
    rng = np.random.default_rng(31)
    src = tmp_path / "in"
    src.mkdir()
    for i, (h, w) in enumerate([(40, 60), (40, 60), (33, 47), (40, 60), (64, 64)]):
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(src / f"img{i}.jpeg", quality=92)
    for enc in ("pillow", "device"):
        random.seed(5); np.random.seed(5)
        n = io_pipeline.run_directory(str(src), str(tmp_path / enc), chunk_images=2, workers=2, encoder=enc)
        assert n == 40
    names = sorted(os.listdir(tmp_path / "pillow"))
    assert names == sorted(os.listdir(tmp_path / "device"))
    for n in names:
        assert (tmp_path / "pillow" / n).read_bytes() == (tmp_path / "device" / n).read_bytes(), n


def test_output_dir_files_come_from_the_device_writer_and_the_images_still_come_back(device, tmp_path, monkeypatch):
    """apply_all_transformations with `output_dir` set (the reference saves every transformed image, transformation.py:159-162):
    the files are the device writer's — byte-identical to Pillow's save — and the returned images are the same as without saving."""
    import random
    from conftest import synth
    from imagetransformations_amd import transformation as T
    imgs = [(Image.fromarray(synth(330 + i, *hw)), f"/d/img_{i}.JPEG") for i, hw in enumerate([(40, 56), (40, 56), (33, 47), (40, 56), (64, 64)])]
    outs = {}
    for mode in ("device", "pillow"):
        d = tmp_path / mode
        d.mkdir()
        monkeypatch.setattr(T, "output_dir", str(d))
        monkeypatch.setenv("IMGXF_SAVE", mode)
        random.seed(8); np.random.seed(8)
        res = T.apply_all_transformations(imgs)
        outs[mode] = ([np.asarray(im) for im in res], {f: (d / f).read_bytes() for f in sorted(os.listdir(d))})
    assert len(outs["device"][0]) == 40 and len(outs["device"][1]) == 40
    assert outs["device"][1].keys() == outs["pillow"][1].keys()
    for f in outs["device"][1]:
        assert outs["device"][1][f] == outs["pillow"][1][f], f
    for a, b in zip(outs["device"][0], outs["pillow"][0]):
        assert np.array_equal(a, b)
