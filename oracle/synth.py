"""Seeded synthetic inputs shared by the golden generator, the tests and the CPU baseline.

KITTI-shaped stand-ins: box-blurred uniform noise for frames,
Monodepth2's normalised intrinsics (datasets/kitti_dataset.py:29-32 scaled per
mono_dataset.py:333-342), stereo_T[0,3] = -/+0.1 (mono_dataset.py:367-373), a 260x300
object patch with an elliptical mask, and a tiny seeded depth network.

Everything is regenerated from the seed, so goldens only store reference OUTPUTS.
Test infrastructure only (see oracle/__init__.py).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

KITTI_CALIB_TEXT = (
    "P0: 7.215377e+02 0.0 6.095593e+02 0.0 0.0 7.215377e+02 1.728540e+02 0.0 0.0 0.0 1.0 0.0\n"
    "P1: 7.215377e+02 0.0 6.095593e+02 -3.875744e+02 0.0 7.215377e+02 1.728540e+02 0.0 0.0 0.0 1.0 0.0\n"
    "P2: 7.215377e+02 0.0 6.095593e+02 4.485728e+01 0.0 7.215377e+02 1.728540e+02 2.163791e-01 0.0 0.0 1.0 2.745884e-03\n"
    "P3: 7.215377e+02 0.0 6.095593e+02 -3.395242e+02 0.0 7.215377e+02 1.728540e+02 2.199936e+00 0.0 0.0 1.0 2.729905e-03\n"
    "R0_rect: 1.0 0.0 0.0 0.0 1.0 0.0 0.0 0.0 1.0\n"
    "Tr_velo_to_cam: 0.0 -1.0 0.0 0.0 0.0 0.0 -1.0 0.0 1.0 0.0 0.0 0.0\n"
    "Tr_imu_to_velo: 1.0 0.0 0.0 0.0 0.0 1.0 0.0 0.0 0.0 0.0 1.0 0.0\n"
)


def kitti_like(B, C, H, W, gen):
    """5x5 box-blurred U[0,1) noise: image-like SSIM statistics."""
    x = torch.rand(B, C, H + 4, W + 4, generator=gen)
    return F.avg_pool2d(x, 5, 1).contiguous()


def smooth_field(B, H, W, gen, lo, hi, k=9):
    x = torch.rand(B, 1, H + k - 1, W + k - 1, generator=gen)
    x = F.avg_pool2d(x, k, 1)
    x = (x - x.amin()) / (x.amax() - x.amin() + 1e-12)
    return (lo + (hi - lo) * x).contiguous()


def make_intrinsics(B, H, W):
    K = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
    K[0, :] *= W
    K[1, :] *= H
    inv_K = np.linalg.pinv(K)
    K = torch.from_numpy(K).unsqueeze(0).repeat(B, 1, 1)
    inv_K = torch.from_numpy(inv_K).unsqueeze(0).repeat(B, 1, 1)
    return K.contiguous(), inv_K.contiguous()


def make_object(seed=7, h=260, w=300):
    g = torch.Generator().manual_seed(seed)
    patch = torch.rand(1, 3, h, w, generator=g)
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    mask = ((((xs - (w - 1) / 2) / 140.0) ** 2 + ((ys - (h - 1) / 2) / 110.0) ** 2) <= 1.0).float()
    return patch.contiguous(), mask.view(1, 1, h, w).contiguous()


def make_loss_case(B, H, W, seed, disp_lo=0.01, disp_hi=0.35, dtype=torch.float32):
    """inputs dict (colour pyramids of the target, right view, K, inv_K, stereo_T) + 4 disparities."""
    g = torch.Generator().manual_seed(seed)
    left = kitti_like(B, 3, H, W, g)
    shift = max(2, W // 128)
    right = (0.9 * torch.roll(left, shift, dims=3) + 0.1 * kitti_like(B, 3, H, W, g)).contiguous()
    inputs = {}
    for s in range(4):
        inputs[("color", 0, s)] = (left if s == 0 else F.avg_pool2d(left, 2 ** s)).contiguous().to(dtype)
    inputs[("color", "s", 0)] = right.to(dtype)
    K, inv_K = make_intrinsics(B, H, W)
    inputs[("K", 0)], inputs[("inv_K", 0)] = K.to(dtype), inv_K.to(dtype)
    T = torch.eye(4).repeat(B, 1, 1)
    for b in range(B):
        T[b, 0, 3] = -0.1 if b % 2 == 0 else 0.1
    inputs["stereo_T"] = T.to(dtype)
    disps = [smooth_field(B, H // 2 ** s, W // 2 ** s, g, disp_lo, disp_hi, k=max(3, 9 // 2 ** s) | 1).to(dtype)
             for s in range(4)]
    return inputs, disps


def add_pyramid(inputs, B, H, W):
    """("color","s",s), ("K",s), ("inv_K",s) for s = 1..3 (MD2/datasets/mono_dataset.py:333-342 scales the normalised
    intrinsics by the size of each scale; the colour pyramid here is an average pool like ("color",0,s) above):
    what --v1_multiscale reads."""
    dtype = inputs[("color", 0, 0)].dtype
    for s in range(1, 4):
        inputs[("color", "s", s)] = F.avg_pool2d(inputs[("color", "s", 0)], 2 ** s).contiguous()
        K, inv_K = make_intrinsics(B, H // 2 ** s, W // 2 ** s)
        inputs[("K", s)], inputs[("inv_K", s)] = K.to(dtype), inv_K.to(dtype)
    return inputs


def make_depth_hint(B, H, W, seed):
    """Synthetic depth hints (DepthHints' SGM stereo estimates, depth-hints/datasets/mono_dataset.py:368-388): a smooth
    depth field in metres-of-the-0.1-baseline units with holes (hint == 0 where no estimate), and its validity mask."""
    g = torch.Generator().manual_seed(seed)
    disp = smooth_field(B, H, W, g, 0.02, 0.3, k=9)
    depth = 1.0 / (0.01 + 9.99 * disp)
    holes = F.avg_pool2d(torch.rand(B, 1, H + 6, W + 6, generator=g), 7, 1) < 0.457      # ~15 % of the pixels, in blobs
    depth = torch.where(holes, torch.zeros_like(depth), depth)
    return depth.contiguous(), (depth > 0).float()


class TinyDepthNet(nn.Module):
    """A seeded 3-layer conv net [B,3,H,W] -> sigmoid disparity [B,1,H,W] with one BatchNorm,
    so that Attack.__call__'s eval()/train() bracket (attack.py:296-312) is observable."""

    def __init__(self, seed=5, ch=8):
        super().__init__()
        self.c1 = nn.Conv2d(3, ch, 3, padding=1)
        self.bn = nn.BatchNorm2d(ch)
        self.c2 = nn.Conv2d(ch, ch, 3, padding=2, dilation=2)
        self.c3 = nn.Conv2d(ch, 1, 3, padding=1)
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for p in self.parameters():
                p.copy_((torch.rand(p.shape, generator=g) - 0.5) * (2.0 / max(1.0, float(p[0].numel())) ** 0.5))
            self.bn.weight.fill_(1.0)
            self.bn.running_mean.copy_((torch.rand(ch, generator=g) - 0.5) * 0.1)
            self.bn.running_var.copy_(torch.rand(ch, generator=g) * 0.5 + 0.75)

    def forward(self, x):
        x = F.elu(self.bn(self.c1((x - 0.45) / 0.225)))
        x = F.elu(self.c2(x))
        return torch.sigmoid(self.c3(x))


def general_pose(B, seed):
    """A small rigid motion per sample (rotation about all three axes + translation), float64 -> float32."""
    g = torch.Generator().manual_seed(seed)
    ang = (torch.rand(B, 3, generator=g, dtype=torch.float64) - 0.5) * 0.02
    T = torch.eye(4, dtype=torch.float64).repeat(B, 1, 1)
    for b in range(B):
        ax, ay, az = [float(v) for v in ang[b]]
        Rx = torch.tensor([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]], dtype=torch.float64)
        Ry = torch.tensor([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]], dtype=torch.float64)
        Rz = torch.tensor([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]], dtype=torch.float64)
        T[b, :3, :3] = Rz @ Ry @ Rx
    T[:, :3, 3] = (torch.rand(B, 3, generator=g, dtype=torch.float64) - 0.5) * 0.06
    return T.float()


def options_case(B, H, W, seed, frames):
    """Inputs of the option-branch goldens: make_loss_case + (for monocular source frames -1 / +1) a shifted, noised copy of
    the target as that frame and a general pose for it; predictive masks = smooth fields in (0.05, 0.95), one channel per
    source frame, at the four scales."""
    inputs, disps = make_loss_case(B, H, W, seed)
    g = torch.Generator().manual_seed(seed + 7)
    poses = {}
    for f in frames:
        if f == "s":
            continue
        inputs[("color", f, 0)] = (0.8 * torch.roll(inputs[("color", 0, 0)], -2 * f, 3) + 0.2 * kitti_like(B, 3, H, W, g)).contiguous()
        poses[f] = general_pose(B, seed + 11 + f)
    masks = [torch.cat([smooth_field(B, H >> s, W >> s, g, 0.05, 0.95, k=max(3, 9 // 2 ** s) | 1) for _ in frames], 1).contiguous()
             for s in range(4)]
    return inputs, disps, poses, masks


def gt_depth_case():
    """Inputs of tests/golden/addon_gt_depth.npz (also regenerated by the tests from these seeds): disparities that reach
    both clamp bounds of the metric depth (5.4 * depth > 80 for disp < 0.0058), a soft-edged object mask, one distance per
    sample."""
    B, H, W = 4, 32, 96
    g = torch.Generator().manual_seed(63)
    color_ben = kitti_like(B, 3, H, W, g)
    disp = torch.rand(B, 1, H, W, generator=g) * 0.3 + 0.01
    disp = torch.where(torch.rand(B, 1, H, W, generator=g) < 0.15, disp * 0.02, disp)      # far pixels: clamped to 80 m
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    cx = torch.tensor([30.0, 48.0, 60.0, 20.0]).view(B, 1, 1)
    r = ((xx - cx) ** 2 / 18.0 ** 2 + (yy - 18.0) ** 2 / 9.0 ** 2).sqrt()
    mask = (1.5 - r).clamp(0, 1).unsqueeze(1).expand(-1, 3, -1, -1).contiguous()            # bilinear-warped masks have soft edges
    objdepth = torch.tensor([5.0, 7.4, 9.8, 6.2]).view(B, 1, 1)
    return color_ben, disp, mask, objdepth


def decoder_case(seed=72, B=2, H=64, W=192):
    """Inputs of tests/golden/unet_decoder.npz: five encoder-shaped feature maps of a H x W frame and one weight map per
    disparity scale (the scalar sum_s <disp_s, w_s> is differentiated)."""
    g = torch.Generator().manual_seed(seed)
    ch = [64, 64, 128, 256, 512]
    feats = [torch.rand(B, c, H >> (k + 1), W >> (k + 1), generator=g) * 2 - 0.5 for k, c in enumerate(ch)]
    wts = [torch.rand(B, 1, H >> s, W >> s, generator=g) - 0.3 for s in range(4)]
    return feats, wts
