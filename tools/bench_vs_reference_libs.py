"""Per-transformation comparison on one 3840x2160 RGB frame batch: the library call the reference
makes (Pillow / SciPy / NumPy, single thread, on this box's host CPU — OpenCV is not installed, so
blur / contrast use the oracle's C port instead) next to the HIP path on resident frames.
usage: python tools/bench_vs_reference_libs.py [frames_on_gpu]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from PIL import Image, ImageEnhance
from scipy import ndimage
from imagetransformations_amd import ops, transformation as T

F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H, W = 2160, 3840
rng = np.random.default_rng(0)
a = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
img = Image.fromarray(a)
dev = torch.device("cuda:0")
batch = torch.from_numpy(np.stack([a] * F)).to(dev)

def cpu_time(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e3           # ms per frame

def gpu_time(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it / F                         # ms per frame

def ref_scale(s):
    nw, nh = int(W * s), int(H * s)
    r = img.resize((nw, nh), Image.Resampling.LANCZOS)
    l, t = (nw - W) // 2, (nh - H) // 2
    return r.crop((l, t, l + W, t + H))

def ref_shear(sh):
    shift = int(np.ceil(sh * H))
    return img.transform((W + shift, H), Image.AFFINE, (1, sh, -shift, 0, 1, 0), Image.BICUBIC, fillcolor=(255, 255, 255))

def ref_background():
    gray = img.convert('L')
    edges = ndimage.sobel(np.array(gray))
    mask = edges > np.percentile(edges, 70)
    fg = ndimage.binary_dilation(mask, iterations=3)
    return Image.composite(img, Image.new('RGB', img.size, (10, 200, 30)), Image.fromarray((fg * 255).astype(np.uint8)))

def persp_coeffs():
    st = [[0, 0], [W - 1, 0], [W - 1, H - 1], [0, H - 1]]
    en = [[300, 150], [W - 200, 90], [W - 350, H - 120], [120, H - 200]]     # a RandomPerspective(0.2) draw
    m = np.zeros((8, 8)); b = np.array(st, float).reshape(8)
    for i, (p1, p2) in enumerate(zip(en, st)):
        m[2 * i] = [p1[0], p1[1], 1, 0, 0, 0, -p2[0] * p1[0], -p2[0] * p1[1]]
        m[2 * i + 1] = [0, 0, 0, p1[0], p1[1], 1, -p2[1] * p1[0], -p2[1] * p1[1]]
    return [float(v) for v in np.linalg.solve(m, b).astype(np.float32)]

PC = persp_coeffs()

def ref_perspective():
    """ToTensor -> F.perspective (tensor path: grid + grid_sample + mask blend) -> ToPILImage, on the
    torch CPU primitives torchvision is made of (torchvision itself is not installed)."""
    import torch.nn.functional as Fn
    t = torch.from_numpy(a).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    th1 = torch.tensor([[[PC[0], PC[1], PC[2]], [PC[3], PC[4], PC[5]]]])
    th2 = torch.tensor([[[PC[6], PC[7], 1.0], [PC[6], PC[7], 1.0]]])
    base = torch.empty(1, H, W, 3)
    base[..., 0].copy_(torch.linspace(0.5, W - 0.5, steps=W))
    base[..., 1].copy_(torch.linspace(0.5, H - 0.5, steps=H).unsqueeze_(-1))
    base[..., 2].fill_(1)
    g1 = base.view(1, H * W, 3).bmm(th1.transpose(1, 2) / torch.tensor([0.5 * W, 0.5 * H]))
    g2 = base.view(1, H * W, 3).bmm(th2.transpose(1, 2))
    grid = (g1 / g2 - 1.0).view(1, H, W, 2)
    im = torch.cat((t.unsqueeze(0), torch.ones(1, 1, H, W)), dim=1)
    im = Fn.grid_sample(im, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    mask = im[:, -1:].expand(1, 3, H, W)
    im = im[:, :-1] * mask + (1.0 - mask) * torch.zeros(1, 3, 1, 1)
    return im.squeeze(0).mul(255).byte()

rows = [
    ("apply_rotation 30 (Image.rotate NEAREST)", lambda: img.rotate(-30, fillcolor=(0, 0, 0)), lambda: T._rotation_t(batch, 30.0)),
    ("rotate 30 + 1.5x BILINEAR (Image.transform)", lambda: img.transform((W, H), Image.AFFINE, ops.rotate_zoom_matrix(W, H, 30.0, 1.5), Image.BILINEAR),
     lambda: ops.affine(batch, ops.rotate_zoom_matrix(W, H, 30.0, 1.5), (W, H), ops.BILINEAR, (0, 0, 0))),
    ("apply_scale 1.1 (resize LANCZOS + crop)", lambda: ref_scale(1.1), lambda: T._scale_t(batch, 1.1)),
    ("apply_shear 0.3 (transform BICUBIC)", lambda: ref_shear(0.3), lambda: T._shear_t(batch, 0.3)),
    ("apply_brightness 0.05 (ImageEnhance)", lambda: ImageEnhance.Brightness(img).enhance(1.05), lambda: ops.brightness(batch, 1.05)),
    ("apply_background_change (L, sobel, percentile, dilation, composite)", ref_background,
     lambda: ops.composite_const(batch, (10, 200, 30), ops.dilate_cross(ops.percentile_mask(ops.rgb_sobel(batch), 70), 3))),
    (f"apply_perspective_warp 0.2 (torchvision tensor path, torch CPU {torch.get_num_threads()} thr)", ref_perspective, lambda: ops.perspective(batch, PC)),
    ("ImageOps.equalize (AugMix)", lambda: __import__("PIL.ImageOps", fromlist=["equalize"]).equalize(img), lambda: ops.equalize(batch)),
]
print(f"{'transformation (library call the reference makes)':72s} {'CPU ms/frame':>12s} {'HIP ms/frame':>12s} {'ratio':>8s}")
for name, cpu, gpu in rows:
    c = cpu_time(cpu); g = gpu_time(gpu)
    print(f"{name:72s} {c:12.2f} {g:12.4f} {c / g:8.0f}x", flush=True)
