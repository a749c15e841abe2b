"""Where does the fused loss's disparity gradient differ from the float64 oracle?  (GPU; diagnostic for the parity tests.)

    python tools/diag_grad_outliers.py B H W seed [scale]

For one seeded case: rel-L2 of HIP and of the fp32 oracle against float64 (all elements, and the [::3, ::3] sub-sample the big
fixtures store), how concentrated each error is, and for HIP's largest outliers the two float64 margins that make a pixel
ill-conditioned in ANY fp32 run: the identity / reprojection gap of the per-pixel min (over the 3x3 neighbourhood that shares
SSIM windows with it) and the distance of the sample coordinate from the next integer (the bilinear floor())."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthmodelhardening_amd import _native as N, ops  # noqa: E402
from oracle import loss_ref, synth  # noqa: E402


def oracle(B, H, W, seed, dtype):
    inputs, disps = synth.make_loss_case(B, H, W, seed, dtype=dtype)
    outputs = {("disp", s): disps[s].clone().requires_grad_(True) for s in range(4)}
    loss_ref.generate_images_pred(inputs, outputs)
    losses, maps = loss_ref.compute_losses(inputs, outputs, noise=None)
    losses["loss"].backward()
    return inputs, disps, outputs


def main():
    B, H, W, seed = [int(v) for v in sys.argv[1:5]]
    s = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    i64, d64, o64 = oracle(B, H, W, seed, torch.float64)
    i32, d32, o32 = oracle(B, H, W, seed, torch.float32)
    dev = torch.device("cuda")
    dd = [d.to(dev).requires_grad_(True) for d in d32]
    out = ops.photometric_smooth_loss(i32[("color", 0, 0)].to(dev), [i32[("color", "s", 0)].to(dev)], [i32["stereo_T"].to(dev)],
                                      i32[("K", 0)].to(dev), i32[("inv_K", 0)].to(dev), dd,
                                      [i32[("color", 0, k)].to(dev) for k in range(4)], noise=None)
    out.fin[N.FIN_LOSS].backward()
    g64 = o64[("disp", s)].grad
    gh, gr = dd[s].grad.double().cpu(), o32[("disp", s)].grad.double()

    def rel(a, b):
        return float((a - b).norm() / b.norm())
    print("scale %d  rel-L2 vs fp64: hip %.3g  ref32 %.3g   | [::3,::3]: hip %.3g  ref32 %.3g" % (
        s, rel(gh, g64), rel(gr, g64), rel(gh[:, :, ::3, ::3], g64[:, :, ::3, ::3]), rel(gr[:, :, ::3, ::3], g64[:, :, ::3, ::3])))
    for name, g in (("hip", gh), ("ref32", gr)):
        e2 = (g - g64).pow(2).flatten()
        top = e2.topk(50).values
        print("%-5s squared error carried by the top 1 / 10 / 50 elements: %.3f %.3f %.3f; elements beyond 1e-4 max|g|: %d" % (
            name, float(top[0] / e2.sum()), float(top[:10].sum() / e2.sum()), float(top.sum() / e2.sum()),
            int(((g - g64).abs() > 1e-4 * g64.abs().max()).sum())))
    # float64 margins at full resolution
    tgt = i64[("color", 0, 0)]
    ident = loss_ref.compute_reprojection_loss(i64[("color", "s", 0)], tgt)
    reproj = loss_ref.compute_reprojection_loss(o64[("color", "s", s)].detach(), tgt)
    gap = (ident - reproj).abs()[:, 0]                                  # [B,H,W]
    gap3 = -F.max_pool2d(-gap.unsqueeze(1), 5, 1, 2)[:, 0]              # smallest gap among the pixels sharing SSIM windows
    grid = o64[("sample", "s", s)].detach()                             # [B,H,W,2] in [-1,1]
    px = (grid[..., 0] + 1) * 0.5 * (W - 1)
    fx = (px - px.floor())
    fdist = torch.minimum(fx, 1 - fx)
    fd3 = -F.max_pool2d(-fdist.unsqueeze(1), 5, 1, 2)[:, 0]
    f = 2 ** s
    err = (gh - g64).abs()
    idx = err.flatten().topk(15).indices
    print("HIP's 15 largest deviations (scale %d): b y x | err / max|g| | g64 | min id/reproj gap and min floor distance over the "
          "full-resolution pixels this texel reaches" % s)
    gmax = float(g64.abs().max())
    hs, ws = H // f, W // f
    for i in idx.tolist():
        b, rem = divmod(i, hs * ws)
        y, x = divmod(rem, ws)
        y0, y1, x0, x1 = max(0, (y - 1) * f), min(H, (y + 2) * f), max(0, (x - 1) * f), min(W, (x + 2) * f)
        print("  %d %4d %4d | %.3g | %.3g | gap %.3g  floor-dist %.3g | ref32 err %.3g" % (
            b, y, x, float(err.flatten()[i]) / gmax, float(g64.flatten()[i]) / gmax, float(gap3[b, y0:y1, x0:x1].min()),
            float(fd3[b, y0:y1, x0:x1].min()), float((gr - g64).abs().flatten()[i]) / gmax))
    print("fraction of pixels with gap < 1e-5: %.3g ; with floor-dist < 1e-4: %.3g" % (
        float((gap < 1e-5).double().mean()), float((fdist < 1e-4).double().mean())))


if __name__ == "__main__":
    main()
