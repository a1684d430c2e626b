"""Time every device op of ops.py once on a 4K RGB batch: a scan for kernels far from the HBM roofline.
usage: python tools/scan_ops.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from imagetransformations_amd import ops
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H, W = 2160, 3840
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
t = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device=dev, generator=g)
gray = ops.rgb2l(t)
mask = (gray > 128).to(torch.uint8) * 255
noise = torch.randn((F, H, W, 3), device=dev, generator=g) * 10
def run(fn, iters=3):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
px = F * H * W
cases = [
    ("flip left-right", lambda: ops.flip(t), 6), ("flip top-bottom", lambda: ops.flip(t, True), 6),
    ("rot90", lambda: ops.rot90(t, 1), 6), ("rot180", lambda: ops.rot90(t, 2), 6),
    ("translate (affine nearest 1,0,-37,0,1,21)", lambda: ops.affine(t, [1, 0, -37, 0, 1, 21], (W, H), ops.NEAREST, (0, 0, 0)), 6),
    ("camera distance (nearest zoom 1.2)", lambda: ops.affine(t, ops.rotate_zoom_matrix(W, H, 0.0, 1.2), (W, H), ops.NEAREST, (0, 0, 0)), 5),
    ("apply_translation (+50, -35)", lambda: ops.translate(t, 50, -35), 6),
    ("crop 3000x2000", lambda: ops.crop(t, (100, 50, 3100, 2050)), 6 * 3000 * 2000 / (H * W)),
    ("box_blur r=2", lambda: ops.box_blur(t, 2.0), 6), ("gaussian_blur_pil r=2 (defocus)", lambda: ops.gaussian_blur_pil(t, 2.0), 6),
    ("gaussian_blur_pil r=6 (defocus severity 3)", lambda: ops.gaussian_blur_pil(t, 6.0), 6), ("gaussian_blur_pil r=10 (severity 5)", lambda: ops.gaussian_blur_pil(t, 10.0), 6),
    ("filter3x3 smooth", lambda: ops.filter3x3(t, [1, 1, 1, 1, 5, 1, 1, 1, 1], 13.0), 6),
    ("enhance_sharpness 1.5", lambda: ops.enhance_sharpness(t, 1.5), 6), ("enhance_color 1.5", lambda: ops.enhance_color(t, 1.5), 6),
    ("enhance_contrast 1.5", lambda: ops.enhance_contrast(t, 1.5), 6),
    ("add_noise f32", lambda: ops.add_noise(t, noise), 18),
    ("posterize 4", lambda: ops.posterize(t, 4), 6), ("solarize", lambda: ops.solarize(t, 128), 6), ("equalize (ImageOps)", lambda: ops.equalize(t), 9),
    ("rgb2yuv", lambda: ops.rgb2yuv(t), 6), ("equalize_hist_cv", lambda: ops.equalize_hist_cv(t, 0), 9),
    ("channel_histogram", lambda: ops.channel_histogram(t), 3), ("permute_channels", lambda: ops.permute_channels(t, (2, 1, 0)), 6),
    ("composite_const", lambda: ops.composite_const(t, (255, 0, 0), mask), 7), ("percentile_mask", lambda: ops.percentile_mask(gray, 80.0), 2),
    ("sobel gray x", lambda: ops.sobel(gray), 2), ("conv2d laplace 3x3 (dense)", lambda: ops.conv2d(t, [[0, -1, 0], [-1, 5, -1], [0, -1, 0]]), 6),
]
for name, fn, bpp in cases:
    try:
        ms = run(fn)
        print(f"{name:44s} {ms:8.3f} ms   {bpp * px / ms / 1e6:8.0f} GB/s  {bpp * px / ms / 1e6 / 8000 * 100:5.1f}% of 8 TB/s", flush=True)
    except Exception as e:
        print(f"{name:44s} failed: {type(e).__name__}: {e}", flush=True)
