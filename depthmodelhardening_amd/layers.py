"""Reference-surface layers (MD2/layers.py): same names and call signatures.

The training hot path does NOT go through these modules -- Trainer.compute_losses calls the fused HIP
kernel (ops.photometric_smooth_loss).  They exist so that code written against the reference's
``layers`` module keeps working (evaluation scripts, notebooks): disp_to_depth :16-25, ConvBlock :106-118,
Conv3x3 :121-136, BackprojectDepth :139-168, Project3D :171-198, upsample :201-204,
get_smooth_loss :207-220, SSIM :223-253.  Buffers are ``nn.Parameter(requires_grad=False)`` exactly as
in the reference so that state_dicts keep the same keys.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def disp_to_depth(disp, min_depth, max_depth):
    """Sigmoid output -> (scaled disparity, depth)."""
    min_disp = 1 / max_depth
    max_disp = 1 / min_depth
    scaled_disp = min_disp + (max_disp - min_disp) * disp
    return scaled_disp, 1 / scaled_disp


class Conv3x3(nn.Module):
    """Reflection- (or zero-) pad by one, then a 3x3 convolution."""

    def __init__(self, in_channels, out_channels, use_refl=True):
        super().__init__()
        self.pad = nn.ReflectionPad2d(1) if use_refl else nn.ZeroPad2d(1)
        self.conv = nn.Conv2d(int(in_channels), int(out_channels), 3)

    def forward(self, x):
        return self.conv(self.pad(x))


class ConvBlock(nn.Module):
    """Conv3x3 followed by ELU."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = Conv3x3(in_channels, out_channels)
        self.nonlin = nn.ELU(inplace=True)

    def forward(self, x):
        return self.nonlin(self.conv(x))


def upsample(x):
    return F.interpolate(x, scale_factor=2, mode="nearest")


class BackprojectDepth(nn.Module):
    """Depth image -> homogeneous camera points [B,4,H*W]."""

    def __init__(self, batch_size, height, width):
        super().__init__()
        self.batch_size, self.height, self.width = batch_size, height, width
        xs, ys = np.meshgrid(range(width), range(height), indexing='xy')
        self.id_coords = nn.Parameter(torch.from_numpy(np.stack([xs, ys], 0).astype(np.float32)), requires_grad=False)
        self.ones = nn.Parameter(torch.ones(batch_size, 1, height * width), requires_grad=False)
        flat = torch.stack([self.id_coords[0].view(-1), self.id_coords[1].view(-1)], 0).unsqueeze(0)
        flat = flat.repeat(batch_size, 1, 1)
        self.pix_coords = nn.Parameter(torch.cat([flat, self.ones], 1), requires_grad=False)

    def forward(self, depth, inv_K):
        cam_points = torch.matmul(inv_K[:, :3, :3], self.pix_coords)
        cam_points = depth.view(self.batch_size, 1, -1) * cam_points
        return torch.cat([cam_points, self.ones], 1)


class Project3D(nn.Module):
    """Camera points -> normalised sampling grid [B,H,W,2] for a camera with intrinsics K at pose T."""

    def __init__(self, batch_size, height, width, eps=1e-7):
        super().__init__()
        self.batch_size, self.height, self.width, self.eps = batch_size, height, width, eps

    def forward(self, points, K, T):
        P = torch.matmul(K, T)[:, :3, :]
        cam_points = torch.matmul(P, points)
        pix = cam_points[:, :2, :] / (cam_points[:, 2, :].unsqueeze(1) + self.eps)
        pix = pix.view(self.batch_size, 2, self.height, self.width).permute(0, 2, 3, 1)
        pix = torch.stack([pix[..., 0] / (self.width - 1), pix[..., 1] / (self.height - 1)], -1)
        return (pix - 0.5) * 2


def get_smooth_loss(disp, img):
    """Edge-aware first-order smoothness of a (normalised) disparity image (MD2/layers.py:207-220).  fp32 CUDA tensors go
    through the registered kernel ``torch.ops.dmh.smooth_loss`` (one pass forward, a four-neighbour gather backward; the
    gradient flows to ``disp`` -- ``img`` is data); anything else takes the reference's formula below."""
    if (disp.is_cuda and disp.dtype == torch.float32 and img.dtype == torch.float32 and disp.dim() == 4 and disp.shape[1] == 1
            and img.dim() == 4 and disp.shape[2] >= 2 and disp.shape[3] >= 2 and not img.requires_grad
            and disp.shape[0] <= 65535 and disp.shape[2] * disp.shape[3] < (1 << 30)):       # the kernel's own grid limits
        from . import library  # noqa: F401  (registers torch.ops.dmh.*)
        return torch.ops.dmh.smooth_loss(disp, img)
    gdx = torch.abs(disp[:, :, :, :-1] - disp[:, :, :, 1:])
    gdy = torch.abs(disp[:, :, :-1, :] - disp[:, :, 1:, :])
    gix = torch.mean(torch.abs(img[:, :, :, :-1] - img[:, :, :, 1:]), 1, keepdim=True)
    giy = torch.mean(torch.abs(img[:, :, :-1, :] - img[:, :, 1:, :]), 1, keepdim=True)
    return (gdx * torch.exp(-gix)).mean() + (gdy * torch.exp(-giy)).mean()


class SSIM(nn.Module):
    """(1 - SSIM)/2 over 3x3 windows with reflection padding, clamped to [0,1]."""

    def __init__(self):
        super().__init__()
        self.pool = nn.AvgPool2d(3, 1)
        self.refl = nn.ReflectionPad2d(1)
        self.C1 = 0.01 ** 2
        self.C2 = 0.03 ** 2

    def forward(self, x, y):
        if x.is_cuda and x.dtype == torch.float32 and y.dtype == torch.float32 and x.dim() == 4 and x.shape == y.shape \
                and x.shape[2] >= 2 and x.shape[3] >= 2 and x.shape[0] * x.shape[1] <= 65535 \
                and x.shape[2] * x.shape[3] < (1 << 30):      # the kernel's own grid limits: larger shapes take the formula
            # the registered kernel (torch.ops.dmh.ssim_map: window sums forward, coefficient-field gathers backward)
            from . import library  # noqa: F401
            return torch.ops.dmh.ssim_map(x, y)
        x, y = self.refl(x), self.refl(y)
        mu_x, mu_y = self.pool(x), self.pool(y)
        sigma_x = self.pool(x ** 2) - mu_x ** 2
        sigma_y = self.pool(y ** 2) - mu_y ** 2
        sigma_xy = self.pool(x * y) - mu_x * mu_y
        n = (2 * mu_x * mu_y + self.C1) * (2 * sigma_xy + self.C2)
        d = (mu_x ** 2 + mu_y ** 2 + self.C1) * (sigma_x + sigma_y + self.C2)
        return torch.clamp((1 - n / d) / 2, 0, 1)
