"""CPU restatement of the photometric-reprojection + SSIM + smoothness loss path.

Reference (all under /root/reference/DepthNetworks/monodepth2 unless prefixed):
  disp_to_depth               layers.py:16-25
  BackprojectDepth.forward    layers.py:139-168
  Project3D.forward           layers.py:171-198
  get_smooth_loss             layers.py:207-220
  SSIM.forward                layers.py:223-253
  generate_images_pred        trainer.py:472-523
  compute_reprojection_loss   trainer.py:525-537
  compute_losses              trainer.py:539-674
  DepthHints compute_losses   ../depth-hints/trainer.py:557-741

Plain differentiable PyTorch; works in fp32 and fp64.  Test infrastructure only
(see oracle/__init__.py).  Pinned by tests/golden/loss_*.npz, which were produced by
running the reference source itself (oracle/make_goldens.py).
"""
import torch
import torch.nn.functional as F

MIN_DEPTH = 0.1      # options.py:69-72
MAX_DEPTH = 100.0    # options.py:73-76
SMOOTH_WT = 1e-3     # options.py:61-64 (--disparity_smoothness)


def disp_to_depth(disp, min_depth=MIN_DEPTH, max_depth=MAX_DEPTH):
    """layers.py:16-25."""
    min_disp = 1 / max_depth
    max_disp = 1 / min_depth
    scaled_disp = min_disp + (max_disp - min_disp) * disp
    depth = 1 / scaled_disp
    return scaled_disp, depth


def backproject(depth, inv_K):
    """layers.py:163-168 -- depth [B,1,H,W], inv_K [B,4,4] -> cam points [B,4,H*W]."""
    B, _, H, W = depth.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=depth.dtype), torch.arange(W, dtype=depth.dtype), indexing="ij")
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(H * W, dtype=depth.dtype)], 0)
    pix = pix.unsqueeze(0).repeat(B, 1, 1)
    cam = torch.matmul(inv_K[:, :3, :3], pix)
    cam = depth.view(B, 1, -1) * cam
    ones = torch.ones(B, 1, H * W, dtype=depth.dtype)
    return torch.cat([cam, ones], 1)


def project3d(points, K, T, H, W, eps=1e-7):
    """layers.py:182-198 -- returns the normalised sampling grid [B,H,W,2]."""
    B = points.shape[0]
    P = torch.matmul(K, T)[:, :3, :]
    cam = torch.matmul(P, points)
    pix = cam[:, :2, :] / (cam[:, 2, :].unsqueeze(1) + eps)
    pix = pix.view(B, 2, H, W).permute(0, 2, 3, 1)
    x = pix[..., 0] / (W - 1)
    y = pix[..., 1] / (H - 1)
    pix = torch.stack([x, y], -1)
    return (pix - 0.5) * 2


def ssim(x, y):
    """layers.py:239-253."""
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    x = F.pad(x, [1, 1, 1, 1], mode="reflect")
    y = F.pad(y, [1, 1, 1, 1], mode="reflect")
    mu_x = F.avg_pool2d(x, 3, 1)
    mu_y = F.avg_pool2d(y, 3, 1)
    sigma_x = F.avg_pool2d(x ** 2, 3, 1) - mu_x ** 2
    sigma_y = F.avg_pool2d(y ** 2, 3, 1) - mu_y ** 2
    sigma_xy = F.avg_pool2d(x * y, 3, 1) - mu_x * mu_y
    n = (2 * mu_x * mu_y + C1) * (2 * sigma_xy + C2)
    d = (mu_x ** 2 + mu_y ** 2 + C1) * (sigma_x + sigma_y + C2)
    return torch.clamp((1 - n / d) / 2, 0, 1)


def get_smooth_loss(disp, img):
    """layers.py:207-220."""
    gdx = torch.abs(disp[:, :, :, :-1] - disp[:, :, :, 1:])
    gdy = torch.abs(disp[:, :, :-1, :] - disp[:, :, 1:, :])
    gix = torch.mean(torch.abs(img[:, :, :, :-1] - img[:, :, :, 1:]), 1, keepdim=True)
    giy = torch.mean(torch.abs(img[:, :, :-1, :] - img[:, :, 1:, :]), 1, keepdim=True)
    gdx = gdx * torch.exp(-gix)
    gdy = gdy * torch.exp(-giy)
    return gdx.mean() + gdy.mean()


def normalised_smooth_loss(disp, color):
    """trainer.py:662-664."""
    mean_disp = disp.mean(2, True).mean(3, True)
    norm_disp = disp / (mean_disp + 1e-7)
    return get_smooth_loss(norm_disp, color)


def compute_reprojection_loss(pred, target, no_ssim=False):
    """trainer.py:525-537."""
    l1 = torch.abs(target - pred).mean(1, True)
    if no_ssim:
        return l1
    return 0.85 * ssim(pred, target).mean(1, True) + 0.15 * l1


def warp_view(disp, source, K, inv_K, T, H, W, min_depth=MIN_DEPTH, max_depth=MAX_DEPTH):
    """One (scale, frame) body of generate_images_pred, trainer.py:481-519.
    Returns (depth [B,1,H,W], sample grid [B,H,W,2], warped colour [B,3,H,W])."""
    disp_up = F.interpolate(disp, [H, W], mode="bilinear", align_corners=False)
    _, depth = disp_to_depth(disp_up, min_depth, max_depth)
    cam = backproject(depth, inv_K)
    grid = project3d(cam, K, T, H, W)
    warped = F.grid_sample(source, grid, padding_mode="border", align_corners=True)
    return depth, grid, warped


def generate_images_pred(inputs, outputs, frame_ids=(0, "s"), scales=(0, 1, 2, 3), H=None, W=None):
    """trainer.py:472-523 (non-v1_multiscale, automasking on, no posecnn)."""
    if H is None:
        H, W = inputs[("color", 0, 0)].shape[-2:]
    for scale in scales:
        for frame_id in frame_ids[1:]:
            T = inputs["stereo_T"] if frame_id == "s" else outputs[("cam_T_cam", 0, frame_id)]
            depth, grid, warped = warp_view(outputs[("disp", scale)], inputs[("color", frame_id, 0)],
                                            inputs[("K", 0)], inputs[("inv_K", 0)], T, H, W)
            outputs[("depth", 0, scale)] = depth
            outputs[("sample", frame_id, scale)] = grid
            outputs[("color", frame_id, scale)] = warped
            outputs[("color_identity", frame_id, scale)] = inputs[("color", frame_id, 0)]


def warp_hint(depth_hint, source, K, inv_K, T, H, W):
    """depth-hints/trainer.py:510-525: the source view warped with the depth HINT.  The reference calls
    F.grid_sample(..., padding_mode="border") without align_corners here, i.e. align_corners=False (the default since
    torch 1.3), unlike the main warp (:500-504)."""
    cam = backproject(depth_hint, inv_K)
    grid = project3d(cam, K, T, H, W)
    return F.grid_sample(source, grid, padding_mode="border", align_corners=False)


def compute_losses(inputs, outputs, frame_ids=(0, "s"), scales=(0, 1, 2, 3), noise=None,
                   smooth_wt=SMOOTH_WT, variant="md2", use_depth_hints=False):
    """trainer.py:588-674 (variant="md2") or depth-hints/trainer.py:638-741
    (variant="dh", no depth hints), photometric + smoothness part only.

    ``noise``: dict scale -> tensor shaped like the identity loss ([B,F,H,W] for md2,
    [B,1,H,W] for dh), the already-scaled tie-break term (reference: randn*1e-5,
    trainer.py:642-645); None -> zeros.
    ``use_depth_hints`` (variant "dh" only; depth-hints/trainer.py:629-636,700-725): inputs["depth_hint"] [B,1,H,W] and
    inputs["depth_hint_mask"]; argmin over [reprojection, identity, hint reprojection], proxy log-L1 supervision where
    the hint wins.  Needs outputs[("depth", 0, s)] (generate_images_pred).
    Returns (losses dict, per-scale dict of to_optimise maps)."""
    losses, maps = {}, {}
    total = 0
    hint_reproj = None
    if use_depth_hints:
        assert variant == "dh" and frame_ids[1:] == ("s",)
        H, W = inputs[("color", 0, 0)].shape[-2:]
        pred = warp_hint(inputs["depth_hint"], inputs[("color", "s", 0)], inputs[("K", 0)], inputs[("inv_K", 0)],
                         inputs["stereo_T"], H, W)
        outputs[("color_depth_hint", "s", 0)] = pred
        hint_reproj = compute_reprojection_loss(pred, inputs[("color", 0, 0)]) + 1000 * (1 - inputs["depth_hint_mask"])
    for scale in scales:
        disp = outputs[("disp", scale)]
        color = inputs[("color", 0, scale)]
        target = inputs[("color", 0, 0)]
        reproj = torch.cat([compute_reprojection_loss(outputs[("color", f, scale)], target)
                            for f in frame_ids[1:]], 1)
        ident = torch.cat([compute_reprojection_loss(inputs[("color", f, 0)], target)
                           for f in frame_ids[1:]], 1)
        if variant == "md2":
            if noise is not None:
                ident = ident + noise[scale]
            combined = torch.cat((ident, reproj), dim=1)
            to_opt, idxs = torch.min(combined, dim=1)
            outputs["identity_selection/{}".format(scale)] = (idxs > ident.shape[1] - 1).float()
            loss = to_opt.mean()
            maps[scale] = to_opt
        else:
            ident, _ = torch.min(ident, dim=1, keepdim=True)
            reproj, _ = torch.min(reproj, dim=1, keepdim=True)
            if noise is not None:
                ident = ident + noise[scale]
            cands = [reproj, ident] + ([hint_reproj] if hint_reproj is not None else [])
            idxs = torch.argmin(torch.cat(cands, dim=1), dim=1, keepdim=True)
            mask = (idxs != 1).float()
            rl = (reproj * mask).sum() / (mask.sum() + 1e-7)
            outputs["identity_selection/{}".format(scale)] = (1 - mask).float()
            losses["reproj_loss/{}".format(scale)] = rl
            loss = rl
            maps[scale] = reproj * mask
            if hint_reproj is not None:
                hmask = (idxs == 2).float()
                pred_depth = outputs[("depth", 0, scale)]
                hl = torch.log(torch.abs(inputs["depth_hint"] - pred_depth) + 1) * inputs["depth_hint_mask"] * hmask
                hl = hl.sum() / (hmask.sum() + 1e-7)
                outputs["depth_hint_pixels/{}".format(scale)] = hmask
                losses["depth_hint_loss/{}".format(scale)] = hl
                loss = loss + hl
        smooth = normalised_smooth_loss(disp, color)
        loss = loss + smooth_wt * smooth / (2 ** scale)
        total = total + loss
        losses["loss/{}".format(scale)] = loss
    losses["loss"] = total / len(scales)
    return losses, maps


def compute_losses_options(inputs, outputs, frame_ids=(0, "s"), scales=(0, 1, 2, 3), noise=None, smooth_wt=SMOOTH_WT,
                           automask=True, avg_reprojection=False, predictive_mask=None, no_ssim=False):
    """The option branches of Monodepth2's per-scale body, trainer.py:589-668 (non-v1_multiscale), that compute_losses above
    leaves out: --disable_automasking (:608 skipped), --avg_reprojection (:617-621, :637-640: the MEAN over the source frames
    replaces the per-frame candidates), --predictive_mask (:623-635: requires automasking off, trainer.py:123-125;
    ``predictive_mask`` = {scale: mask [B, F, H/2^s, W/2^s] in (0,1)} as the second decoder returns it, up-sampled to the
    frame, multiplies the reprojection losses, and 0.2 * BCE(mask, 1) is added to the scale's loss).
    ``noise``: {scale: tensor shaped like the identity loss}, already scaled (trainer.py:642-645); None -> zeros.
    Needs outputs[("color", f, s)] (generate_images_pred).  Returns (losses dict, per-scale to_optimise maps)."""
    losses, maps = {}, {}
    total = 0
    H, W = inputs[("color", 0, 0)].shape[-2:]
    for scale in scales:
        loss = 0
        disp = outputs[("disp", scale)]
        color = inputs[("color", 0, scale)]
        target = inputs[("color", 0, 0)]
        reproj = torch.cat([compute_reprojection_loss(outputs[("color", f, scale)], target, no_ssim)
                            for f in frame_ids[1:]], 1)
        ident = None
        if automask:
            ident = torch.cat([compute_reprojection_loss(inputs[("color", f, 0)], target, no_ssim) for f in frame_ids[1:]], 1)
            if avg_reprojection:
                ident = ident.mean(1, keepdim=True)
        elif predictive_mask is not None:
            mask = F.interpolate(predictive_mask[scale], [H, W], mode="bilinear", align_corners=False)
            reproj = reproj * mask
            loss = loss + 0.2 * F.binary_cross_entropy(mask, torch.ones_like(mask))
        if avg_reprojection:
            reproj = reproj.mean(1, keepdim=True)
        if automask:
            if noise is not None:
                ident = ident + noise[scale]
            combined = torch.cat((ident, reproj), dim=1)
        else:
            combined = reproj
        if combined.shape[1] == 1:
            to_opt = combined
        else:
            to_opt, idxs = torch.min(combined, dim=1)
        if automask:
            outputs["identity_selection/{}".format(scale)] = (idxs > ident.shape[1] - 1).float()
        loss = loss + to_opt.mean()
        maps[scale] = to_opt
        loss = loss + smooth_wt * normalised_smooth_loss(disp, color) / (2 ** scale)
        total = total + loss
        losses["loss/{}".format(scale)] = loss
    losses["loss"] = total / len(scales)
    return losses, maps


def compute_losses_options_dh(inputs, outputs, frame_ids=(0, "s"), scales=(0, 1, 2, 3), noise=None, smooth_wt=SMOOTH_WT,
                              automask=True, avg_reprojection=False, predictive_mask=None, use_depth_hints=False):
    """The same option branches in DepthHints' per-scale body, depth-hints/trainer.py:638-741: the candidates are reduced over
    the source frames FIRST (minimum "as we go", :668-671 / :693-697, or the mean with --avg_reprojection), the tie-break noise
    has ONE channel (:699-702), the masks come from compute_loss_masks (:559-590: without auto-masking every pixel counts) and the
    scale's photometric term is sum(loss * mask) / (sum(mask) + 1e-7) (:712-713); --predictive_mask multiplies the per-frame
    losses before that reduction and adds 0.2 * BCE(mask, 1) (:674-687).  ``use_depth_hints`` (needs the stereo frame and
    auto-masking: without it compute_loss_masks evaluates ``if <tensor>:`` and raises, :568) adds the hint candidate and the
    proxy log-L1 term (:629-636, :716-727).  Needs outputs[("color", f, s)] and ("depth", 0, s) (generate_images_pred)."""
    losses, maps = {}, {}
    total = 0
    H, W = inputs[("color", 0, 0)].shape[-2:]
    target = inputs[("color", 0, 0)]
    hint_reproj = None
    if use_depth_hints:
        assert automask and "s" in frame_ids[1:]
        pred = warp_hint(inputs["depth_hint"], inputs[("color", "s", 0)], inputs[("K", 0)], inputs[("inv_K", 0)],
                         inputs["stereo_T"], H, W)
        outputs[("color_depth_hint", "s", 0)] = pred
        hint_reproj = compute_reprojection_loss(pred, target) + 1000 * (1 - inputs["depth_hint_mask"])
    for scale in scales:
        loss = 0
        disp = outputs[("disp", scale)]
        color = inputs[("color", 0, scale)]
        reproj = torch.cat([compute_reprojection_loss(outputs[("color", f, scale)], target) for f in frame_ids[1:]], 1)
        ident = None
        if automask:
            ident = torch.cat([compute_reprojection_loss(inputs[("color", f, 0)], target) for f in frame_ids[1:]], 1)
            ident = ident.mean(1, keepdim=True) if avg_reprojection else torch.min(ident, dim=1, keepdim=True)[0]
        elif predictive_mask is not None:
            mask = F.interpolate(predictive_mask[scale], [H, W], mode="bilinear", align_corners=False)
            reproj = reproj * mask
            loss = loss + 0.2 * F.binary_cross_entropy(mask, torch.ones_like(mask))
        reproj = reproj.mean(1, keepdim=True) if avg_reprojection else torch.min(reproj, dim=1, keepdim=True)[0]
        if automask:
            if noise is not None:
                ident = ident + noise[scale]
            cands = [reproj, ident] + ([hint_reproj] if hint_reproj is not None else [])
            idxs = torch.argmin(torch.cat(cands, dim=1), dim=1, keepdim=True)
            rmask = (idxs != 1).float()
        else:
            rmask = torch.ones_like(reproj)
        rl = (reproj * rmask).sum() / (rmask.sum() + 1e-7)
        outputs["identity_selection/{}".format(scale)] = (1 - rmask).float()
        losses["reproj_loss/{}".format(scale)] = rl
        maps[scale] = reproj * rmask
        loss = loss + rl
        if hint_reproj is not None:
            hmask = (idxs == 2).float()
            hl = torch.log(torch.abs(inputs["depth_hint"] - outputs[("depth", 0, scale)]) + 1) * inputs["depth_hint_mask"] * hmask
            hl = hl.sum() / (hmask.sum() + 1e-7)
            outputs["depth_hint_pixels/{}".format(scale)] = hmask
            losses["depth_hint_loss/{}".format(scale)] = hl
            loss = loss + hl
        loss = loss + smooth_wt * normalised_smooth_loss(disp, color) / (2 ** scale)
        total = total + loss
        losses["loss/{}".format(scale)] = loss
    losses["loss"] = total / len(scales)
    return losses, maps


def v1_multiscale_losses(inputs, disps, noise=None, smooth_wt=SMOOTH_WT):
    """--v1_multiscale (trainer.py:478-483,593-596): every scale is warped, compared and smoothed at its own resolution
    with the intrinsics of that scale; loss = mean_s(loss_s), loss_s = mean(min(identity, reprojection)) + wt *
    smooth_s / 2^s.  ``disps``: four leaf tensors.  Returns (losses dict, per-scale outputs dicts)."""
    losses, outs = {}, []
    total = 0
    for s in range(4):
        sub = {("color", 0, 0): inputs[("color", 0, s)], ("color", "s", 0): inputs[("color", "s", s)],
               ("K", 0): inputs[("K", s)], ("inv_K", 0): inputs[("inv_K", s)], "stereo_T": inputs["stereo_T"]}
        o = {("disp", 0): disps[s]}
        generate_images_pred(sub, o, scales=(0,))
        ls, _ = compute_losses(sub, o, scales=(0,), noise=None if noise is None else {0: noise[s]},
                               smooth_wt=smooth_wt / (2 ** s))
        losses["loss/%d" % s] = ls["loss/0"]
        total = total + ls["loss/0"]
        outs.append(o)
    losses["loss"] = total / 4
    return losses, outs
