"""Test helper: torchvision's tensor-path perspective (transforms.functional.perspective on a float
tensor, as RandomPerspective calls it) written out on the torch primitives it is made of
(linspace, bmm, grid_sample, eager arithmetic).  torchvision itself is not installed; torch's
CPU ops are, and they are the arithmetic the reference's apply_perspective_warp
(fall_2025/transformations_code:54-66) ends up in.  Used only to pin oracle.perspective_warp."""
import numpy as np
import torch
import torch.nn.functional as F


def coeffs(startpoints, endpoints):
    a = torch.zeros(2 * len(startpoints), 8, dtype=torch.float64)
    for i, (p1, p2) in enumerate(zip(endpoints, startpoints)):
        a[2 * i, :] = torch.tensor([p1[0], p1[1], 1, 0, 0, 0, -p2[0] * p1[0], -p2[0] * p1[1]], dtype=torch.float64)
        a[2 * i + 1, :] = torch.tensor([0, 0, 0, p1[0], p1[1], 1, -p2[1] * p1[0], -p2[1] * p1[1]], dtype=torch.float64)
    b = torch.tensor(startpoints, dtype=torch.float64).view(8)
    return torch.linalg.lstsq(a, b, driver="gels").solution.to(torch.float32).tolist()


def grid(c, ow, oh):
    dt = torch.float32
    theta1 = torch.tensor([[[c[0], c[1], c[2]], [c[3], c[4], c[5]]]], dtype=dt)
    theta2 = torch.tensor([[[c[6], c[7], 1.0], [c[6], c[7], 1.0]]], dtype=dt)
    d = 0.5
    base = torch.empty(1, oh, ow, 3, dtype=dt)
    base[..., 0].copy_(torch.linspace(d, ow * 1.0 + d - 1.0, steps=ow))
    base[..., 1].copy_(torch.linspace(d, oh * 1.0 + d - 1.0, steps=oh).unsqueeze_(-1))
    base[..., 2].fill_(1)
    rescaled = theta1.transpose(1, 2) / torch.tensor([0.5 * ow, 0.5 * oh], dtype=dt)
    g1 = base.view(1, oh * ow, 3).bmm(rescaled)
    g2 = base.view(1, oh * ow, 3).bmm(theta2.transpose(1, 2))
    return (g1 / g2 - 1.0).view(1, oh, ow, 2)


def perspective_u8(img_u8, c):
    """HWC (or HW) uint8 -> ToTensor -> perspective(BILINEAR, fill=[0]*C) -> ToPILImage bytes."""
    a = img_u8 if img_u8.ndim == 3 else img_u8[..., None]
    h, w, ch = a.shape
    t = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    img = torch.cat((t.unsqueeze(0), torch.ones((1, 1, h, w), dtype=torch.float32)), dim=1)
    img = F.grid_sample(img, grid(c, w, h), mode="bilinear", padding_mode="zeros", align_corners=False)
    mask = img[:, -1:, :, :].expand(1, ch, h, w)
    img = img[:, :-1, :, :]
    fill = torch.tensor([0.0] * ch, dtype=torch.float32).view(1, ch, 1, 1).expand_as(img)
    img = img * mask + (1.0 - mask) * fill
    out = img.squeeze(0).mul(255).byte().permute(1, 2, 0).numpy()
    return out if img_u8.ndim == 3 else out[..., 0]
