"""Restatement of the torchvision==0.8.2 *tensor* ops the hot path calls.

PARITY UNPINNED: torchvision 0.8.2 (reference requirements.txt:93) is a
third-party dependency that is neither vendored under /root/reference nor
installed in this image, and the reference holds no golden vectors for it.  The
functions below restate the published 0.8.2 algorithm
(torchvision/transforms/functional.py + functional_tensor.py at tag v0.8.2) and
are anchored on the reference call sites:

* ``Pad``        physicalTrans.py:114-116
* ``perspective``physicalTrans.py:141-142,160-161,190-191
* ``Resize``     torchattacks/attacks/phy_obj_atk.py:51,89-90,116,120-121,
                 phy_obj_atk_l0.py:34-35,118-119,167,171-172, pgd_depth.py:45

Test infrastructure only (see oracle/__init__.py).
"""
import numpy as np
import torch
import torch.nn.functional as F


def pad(img, padding):
    """transforms.Pad([left, top, right, bottom]) on a tensor: constant zero fill."""
    left, top, right, bottom = [int(p) for p in padding]
    return F.pad(img, [left, right, top, bottom], mode="constant", value=0.0)


def resize(img, size):
    """transforms.Resize(size) on a float tensor in 0.8.2: bilinear,
    align_corners=False, **no antialias** (antialias only arrived in 0.10)."""
    return F.interpolate(img, size=[int(size[0]), int(size[1])], mode="bilinear", align_corners=False)


def get_perspective_coeffs(startpoints, endpoints):
    """functional._get_perspective_coeffs: the 8 coefficients (a..h) mapping an OUTPUT
    pixel (x, y) to the INPUT pixel ((ax+by+c)/(gx+hy+1), (dx+ey+f)/(gx+hy+1)).

    0.8.2 solves the 8x8 system with fp32 ``torch.lstsq`` (LAPACK gels; removed from
    torch since).  The system is square and non-singular for a proper quad, so we solve
    it in float64 and round to fp32 -- what an exact fp32 solver would converge to.
    """
    a = np.zeros((8, 8), dtype=np.float64)
    for i, (p1, p2) in enumerate(zip(endpoints, startpoints)):
        a[2 * i, :] = [p1[0], p1[1], 1, 0, 0, 0, -p2[0] * p1[0], -p2[0] * p1[1]]
        a[2 * i + 1, :] = [0, 0, 0, p1[0], p1[1], 1, -p2[1] * p1[0], -p2[1] * p1[1]]
    b = np.asarray(startpoints, dtype=np.float64).reshape(8)
    res = np.linalg.solve(a, b)
    return [float(np.float32(v)) for v in res]


def perspective_grid(coeffs, ow, oh, dtype, device):
    """functional_tensor._perspective_grid (v0.8.2), op for op."""
    theta1 = torch.tensor([[[coeffs[0], coeffs[1], coeffs[2]],
                            [coeffs[3], coeffs[4], coeffs[5]]]], dtype=dtype, device=device)
    theta2 = torch.tensor([[[coeffs[6], coeffs[7], 1.0],
                            [coeffs[6], coeffs[7], 1.0]]], dtype=dtype, device=device)
    d = 0.5
    base_grid = torch.empty(1, oh, ow, 3, dtype=dtype, device=device)
    base_grid[..., 0].copy_(torch.linspace(d, ow * 1.0 + d - 1.0, steps=ow))
    base_grid[..., 1].copy_(torch.linspace(d, oh * 1.0 + d - 1.0, steps=oh).unsqueeze_(-1))
    base_grid[..., 2].fill_(1)
    rescaled_theta1 = theta1.transpose(1, 2) / torch.tensor([0.5 * ow, 0.5 * oh], dtype=dtype, device=device)
    output_grid1 = base_grid.view(1, oh * ow, 3).bmm(rescaled_theta1)
    output_grid2 = base_grid.view(1, oh * ow, 3).bmm(theta2.transpose(1, 2))
    output_grid = output_grid1 / output_grid2 - 1.0
    return output_grid.view(1, oh, ow, 2)


def perspective_coeffs(img, coeffs):
    """functional_tensor.perspective: bilinear grid_sample, zeros padding,
    align_corners=False (0.8.2 has no ``fill`` for tensors)."""
    ow, oh = img.shape[-1], img.shape[-2]
    grid = perspective_grid(coeffs, ow, oh, img.dtype, img.device)
    if img.shape[0] > 1:
        grid = grid.expand(img.shape[0], grid.shape[1], grid.shape[2], grid.shape[3])
    return F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=False)


def perspective(img, startpoints, endpoints, interpolation=2, fill=None):
    """functional.perspective(img, startpoints, endpoints) for a [N,C,H,W] tensor."""
    coeffs = get_perspective_coeffs([list(map(float, p)) for p in startpoints],
                                    [list(map(float, p)) for p in endpoints])
    return perspective_coeffs(img, coeffs)


# ------------------------------------------------------------------------------------------------------ ColorJitter
# phy_obj_atk_l0.py:41 builds ``self.color_aug = ColorJitter.get_params((0.8, 1.2), (0.8, 1.2), (0.8, 1.2), (-0.1, 0.1))`` ONCE,
# in the constructor, and :122-124 applies it to the pasted scenes when ``color_jit`` is set.  In 0.8.2 get_params draws the
# four factors with ``random.uniform`` (brightness, contrast, saturation, hue, in this order), wraps each adjust_* call in a
# Lambda, ``random.shuffle``s the list and returns the Compose -- a callable, which is what the reference stores (later
# versions return a tuple).  The tensor ops are functional_tensor.py's (v0.8.2), restated op for op below.  PARITY UNPINNED
# like the rest of this file.
def _blend(img1, img2, ratio):
    return (ratio * img1 + (1.0 - ratio) * img2).clamp(0, 1.0)


def rgb_to_grayscale(img):
    r, g, b = img.unbind(dim=-3)
    return (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(dim=-3)


def adjust_brightness(img, factor):
    return _blend(img, torch.zeros_like(img), factor)


def adjust_contrast(img, factor):
    mean = torch.mean(rgb_to_grayscale(img), dim=(-3, -2, -1), keepdim=True)
    return _blend(img, mean, factor)


def adjust_saturation(img, factor):
    return _blend(img, rgb_to_grayscale(img), factor)


def _rgb2hsv(img):
    r, g, b = img.unbind(dim=-3)
    maxc = torch.max(img, dim=-3).values
    minc = torch.min(img, dim=-3).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    cr_divisor = torch.where(eqc, ones, cr)
    rc = (maxc - r) / cr_divisor
    gc = (maxc - g) / cr_divisor
    bc = (maxc - b) / cr_divisor
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = (hr + hg + hb)
    h = torch.fmod((h / 6.0 + 1.0), 1.0)
    return torch.stack((h, s, maxc), dim=-3)


def _hsv2rgb(img):
    h, s, v = img.unbind(dim=-3)
    i = torch.floor(h * 6.0)
    f = (h * 6.0) - i
    i = i.to(dtype=torch.int32)
    p = torch.clamp((v * (1.0 - s)), 0.0, 1.0)
    q = torch.clamp((v * (1.0 - s * f)), 0.0, 1.0)
    t = torch.clamp((v * (1.0 - (s * (1.0 - f)))), 0.0, 1.0)
    i = i % 6
    mask = i.unsqueeze(dim=-3) == torch.arange(6, device=i.device).view(-1, 1, 1)
    a1 = torch.stack((v, q, p, p, t, v), dim=-3)
    a2 = torch.stack((t, v, v, q, p, p), dim=-3)
    a3 = torch.stack((p, p, t, v, v, q), dim=-3)
    a4 = torch.stack((a1, a2, a3), dim=-4)
    return torch.einsum("...ijk, ...xijk -> ...xjk", mask.to(dtype=img.dtype), a4)


def adjust_hue(img, hue_factor):
    if not (-0.5 <= hue_factor <= 0.5):
        raise ValueError('hue_factor ({}) is not in [-0.5, 0.5].'.format(hue_factor))
    h, s, v = _rgb2hsv(img).unbind(dim=-3)
    h = (h + hue_factor) % 1.0
    return _hsv2rgb(torch.stack((h, s, v), dim=-3))


def color_jitter_get_params(brightness, contrast, saturation, hue):
    """transforms.ColorJitter.get_params (v0.8.2): the composed callable.  Draws from ``random`` in the published order."""
    import random
    transforms = []
    if brightness is not None:
        bf = random.uniform(brightness[0], brightness[1])
        transforms.append(lambda img: adjust_brightness(img, bf))
    if contrast is not None:
        cf = random.uniform(contrast[0], contrast[1])
        transforms.append(lambda img: adjust_contrast(img, cf))
    if saturation is not None:
        sf = random.uniform(saturation[0], saturation[1])
        transforms.append(lambda img: adjust_saturation(img, sf))
    if hue is not None:
        hf = random.uniform(hue[0], hue[1])
        transforms.append(lambda img: adjust_hue(img, hf))
    random.shuffle(transforms)

    def compose(img):
        for t in transforms:
            img = t(img)
        return img
    return compose
