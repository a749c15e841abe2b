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
