"""CPU restatement of the physical-object attack inner loops and the EOT paste.

Reference (under /root/reference):
  PhysicalTrans                physicalTrans.py:11-196
  Calibration.project_rect_to_image  preprocessing/kitti_util.py:139-147
  Phy_obj_atk.forward          torchattacks/attacks/phy_obj_atk.py:59-123
  Phy_obj_atk_l0.forward       torchattacks/attacks/phy_obj_atk_l0.py:54-174 (cal_l0 :43-52)
  PGD_depth.forward            torchattacks/attacks/pgd_depth.py:41-80
  Attack.__call__              torchattacks/attack.py:296-320

Plain PyTorch on the CPU.  Random draws use the same global generators the
reference uses (``random.sample``, ``torch`` default generator, ``numpy.random``) in
the same order, so that seeding them reproduces a reference run draw for draw.
Test infrastructure only (see oracle/__init__.py).  Pinned by
tests/golden/atk_*.npz (oracle/make_goldens.py ran the reference loops themselves,
with oracle/tv082.py standing in for the absent torchvision -- see that file).
"""
import random
from math import cos, sin, radians

import numpy as np
import torch
import torch.nn as nn

from . import tv082

ORI_H, ORI_W = 375, 1242                      # my_utils.py:12-13
SCENE_SIZE = [320, 1024]                      # phy_obj_atk.py:50
TRAIN_DIST_RANGE = list(np.arange(5, 10, 0.2))  # my_utils.py:14
ANGLE_RANGE = list(range(-30, 31, 5))         # physicalTrans.py:13

# KITTI object calib 003086 P2 (rows quoted in physicalTrans.py:208-213).
KITTI_P2 = np.array([[7.215377e+02, 0.0, 6.095593e+02, 4.485728e+01],
                     [0.0, 7.215377e+02, 1.728540e+02, 2.163791e-01],
                     [0.0, 0.0, 1.0, 2.745884e-03]], dtype=np.float64)


class PhysicalTransRef(object):
    """physicalTrans.py:11-196 with the calibration passed as the 3x4 P2 matrix."""

    def __init__(self, obj_img, obj_mask, P2=KITTI_P2, output_size=(1, 3, ORI_H, ORI_W),
                 angle_range=None, dist_range=None):
        self.obj_img, self.obj_mask = obj_img, obj_mask
        self.P = np.asarray(P2, dtype=np.float64).reshape(3, 4)
        self.dist_range = list(range(5, 10, 2)) if dist_range is None else dist_range
        self.angle_range = list(ANGLE_RANGE) if angle_range is None else angle_range
        self.output_size = output_size
        assert output_size[2] == ORI_H and output_size[3] == ORI_W
        self.padding_img()
        veh_h, veh_w, cam_h = 1.6, 1.82, 1.65
        self.x0, self.y0, self.m, self.n = 0, cam_h - veh_h / 2, veh_w, veh_h

    def from_za_to_coord(self, z0, alpha):
        x_off = cos(radians(alpha)) * self.m / 2
        z_off = sin(radians(alpha)) * self.m / 2
        x1, x2 = self.x0 - x_off, self.x0 + x_off
        zl, zr = z0 - z_off, z0 + z_off
        y1, y2 = self.y0 - self.n / 2, self.y0 + self.n / 2
        return np.array([[x1, y1, zl], [x2, y1, zr], [x2, y2, zr], [x1, y2, zl]])

    def _project_rect_to_image(self, pts):
        hom = np.hstack((pts, np.ones((pts.shape[0], 1))))
        p2d = np.dot(hom, self.P.T)
        p2d[:, 0] /= p2d[:, 2]
        p2d[:, 1] /= p2d[:, 2]
        return p2d[:, 0:2]

    def obj_pos_on_image(self, z0, alpha, K=None, T=None):
        """objPosOnImage (:62-81) / the point maths of project_w_trans (:178-189)."""
        world = self.from_za_to_coord(z0, alpha)
        n = world.shape[0]
        points = np.concatenate((world.T, np.ones((1, n))), axis=0)
        if K is not None:
            P = (K if T is None else np.matmul(K, T))[:3, :]
            cam = np.matmul(P, points)
            pix = cam[:2, :] / (cam[[2], :] + 1e-7)
            return pix.T.astype(np.int32)
        if T is not None:
            world = np.matmul(T, points).T[:, :3]
        return self._project_rect_to_image(world).astype(np.int32)

    def padding_img(self):
        _, _, H, W = self.obj_img.size()
        _, _, H_out, W_out = self.output_size
        l_pad = (W_out - W) // 2
        r_pad = W_out - W - l_pad
        t_pad = (H_out - H) // 2
        b_pad = H_out - H - t_pad
        self.obj_img_pad = tv082.pad(self.obj_img, [l_pad, t_pad, r_pad, b_pad])
        self.obj_mask_pad = tv082.pad(self.obj_mask, [l_pad, t_pad, r_pad, b_pad])
        self.pos_obj_img_start = [[l_pad, t_pad], [l_pad + W, t_pad], [l_pad + W, t_pad + H], [l_pad, t_pad + H]]

    def reset_img(self, obj_img, obj_mask):
        self.obj_img, self.obj_mask = obj_img, obj_mask
        self.padding_img()

    def project(self, batch_size=1, z0_sample=None, alpha_sample=None, K=None, T=None):
        if z0_sample is None:
            z0_sample = random.sample(self.dist_range, batch_size)
        if alpha_sample is None:
            alpha_sample = random.sample(self.angle_range, batch_size)
        imgs, masks = [], []
        for i in range(batch_size):
            pos = self.obj_pos_on_image(z0_sample[i], alpha_sample[i], K, T)
            imgs.append(tv082.perspective(self.obj_img_pad, self.pos_obj_img_start, pos))
            masks.append(tv082.perspective(self.obj_mask_pad, self.pos_obj_img_start, pos))
        return torch.cat(imgs, 0), torch.cat(masks, 0), z0_sample, alpha_sample


def _tile_scene(images, batch_size):
    if images.size(0) == 1:
        return torch.cat(batch_size * [images.clone()], dim=0)
    if images.size(0) == batch_size:
        return images
    raise RuntimeError("Batch size doesn't match!")


def paste(scene_imgs, trans, batch_size, z0_sample=None, alpha_sample=None):
    """phy_obj_atk.py:87-90: project + composite + resize.  Returns (adv 320x1024, mask 320x1024,
    full-size mask, z0, alpha)."""
    obj, msk, z0, al = trans.project(batch_size=batch_size, z0_sample=z0_sample, alpha_sample=alpha_sample)
    adv = scene_imgs * (1 - msk) + obj * msk
    return tv082.resize(adv, SCENE_SIZE), tv082.resize(msk, SCENE_SIZE), msk, z0, al


def phy_obj_atk(model, obj_img, obj_mask, images, batch_size, eps=0.3, alpha=2 / 255, steps=40,
                random_start=True, dist_range=None, eval=False, P2=KITTI_P2, start_noise=None, record=None,
                draws=None, final_draw=None, trace=None):
    """Phy_obj_atk.forward (phy_obj_atk.py:59-123) wrapped as Attack.__call__ does
    (attack.py:296-312: model.eval() during the attack, train mode restored).
    ``record``: optional list that receives the patch after every step; ``trace``: one that receives (cost, patch gradient).
    ``draws`` / ``final_draw``: explicit (z0_sample, alpha_sample) per step / for the returned scenes, handed to
    project() through its own z0_sample / alpha_sample arguments (physicalTrans.py:130,146-155) instead of its
    ``random.sample`` -- used for batches beyond the 13 poses one draw without replacement can give."""
    dist_range = list(range(5, 31, 2)) if dist_range is None else dist_range
    given_training = model.training
    model.eval()
    trans_adv = PhysicalTransRef(obj_img.clone(), obj_mask, P2, dist_range=dist_range)
    trans_ben = PhysicalTransRef(obj_img, obj_mask, P2, dist_range=dist_range)
    scene_imgs = _tile_scene(images.detach(), batch_size)
    loss = nn.MSELoss()
    adv = obj_img.clone().detach()
    if random_start:
        noise = torch.empty_like(adv).uniform_(-eps, eps) if start_noise is None else start_noise
        adv = torch.clamp(adv + noise, min=0, max=1).detach()
    target = torch.zeros((batch_size, 1, SCENE_SIZE[0], SCENE_SIZE[1]), dtype=obj_img.dtype)
    for step in range(steps):
        adv.requires_grad_()
        trans_adv.reset_img(adv, obj_mask)
        z0_i, al_i = draws[step] if draws is not None else (None, None)
        adv_scenes, masks, _, _, _ = paste(scene_imgs, trans_adv, batch_size, z0_i, al_i)
        cost = -loss(model(adv_scenes) * masks, target)
        grad = torch.autograd.grad(cost, adv, retain_graph=False, create_graph=False)[0]
        if trace is not None:
            trace.append((float(cost), grad.detach().clone()))
        with torch.no_grad():
            adv = adv + alpha * grad.sign()
            delta = torch.clamp(adv - obj_img, min=-eps, max=eps)
            adv = torch.clamp(obj_img + delta, min=0, max=1)
        if record is not None:
            record.append(adv.detach().clone())
    trans_adv.reset_img(adv, obj_mask)
    if final_draw is not None:
        z0, al = list(final_draw[0]), list(final_draw[1])
    else:
        z0 = random.sample(trans_ben.dist_range, batch_size)
        al = random.sample(trans_ben.angle_range, batch_size)
    if eval:
        z0[0], al[0] = 7, 0
    adv_scenes, _, full_mask, _, _ = paste(scene_imgs, trans_adv, batch_size, z0, al)
    obj_ben, _, _, _ = trans_ben.project(batch_size=batch_size, z0_sample=z0, alpha_sample=al)
    ben_scenes = tv082.resize(scene_imgs * (1 - full_mask) + obj_ben * full_mask, SCENE_SIZE)
    masks_out = tv082.resize(full_mask, SCENE_SIZE)
    if given_training:
        model.train()
    return adv_scenes, ben_scenes, masks_out, adv


def cal_l0(pattern_pos, pattern_neg, l0_clip):
    """phy_obj_atk_l0.py:43-52."""
    pp = pattern_pos.detach().clone()
    pn = pattern_neg.detach().clone()
    pp[pp < l0_clip] = 0
    pn[pn > -l0_clip] = 0
    return torch.count_nonzero(torch.sum(torch.abs(pp + pn), dim=1))


def l0_mask_cost(pos_t, neg_t):
    """phy_obj_atk_l0.py:130-132."""
    mp = torch.max(torch.tanh(pos_t / 10) / (2 - 1e-7) + 0.5, dim=1)[0]
    mn = torch.max(torch.tanh(neg_t / 10) / (2 - 1e-7) + 0.5, dim=1)[0]
    return torch.mean(mp) + torch.mean(mn)


def phy_obj_atk_l0(model, obj_img, obj_mask, images, batch_size, adam_lr=0.5, steps=10, mask_wt=0.1,
                   l0_thresh=0.1, dist_range=None, eval=False, P2=KITTI_P2, record=None, color_aug=None):
    """Phy_obj_atk_l0.forward (phy_obj_atk_l0.py:54-174) under Attack.__call__'s eval-mode bracket.  ``color_aug``: the
    callable the constructor got from ColorJitter.get_params (:41; tv082.color_jitter_get_params) when ``color_jit`` is set
    (:122-124), None otherwise."""
    dist_range = list(range(5, 31, 2)) if dist_range is None else dist_range
    clip_max = 1
    l0_clip = clip_max / 255.0
    given_training = model.training
    model.eval()
    obj_img = obj_img.clone().detach()
    obj_mask = obj_mask.clone().detach()
    trans_adv = PhysicalTransRef(obj_img.clone(), obj_mask, P2, dist_range=dist_range)
    trans_ben = PhysicalTransRef(obj_img, obj_mask, P2, dist_range=dist_range)
    scene_imgs = _tile_scene(images.detach(), batch_size)
    pats = []
    for _ in range(2):
        init = np.random.random(obj_img.size()) * clip_max
        init = np.clip(init, 0.0, clip_max) / clip_max
        t = torch.Tensor(init)
        t.requires_grad = True
        pats.append(t)
    pos_t, neg_t = pats
    loss = nn.MSELoss()
    opt = torch.optim.Adam([pos_t, neg_t], lr=adam_lr, betas=(0.5, 0.9))
    target = torch.zeros((batch_size, 1, SCENE_SIZE[0], SCENE_SIZE[1]))
    l0_init = None
    for stp in range(steps * 2):
        p_pos = torch.clamp(pos_t * clip_max, min=0.0, max=clip_max)
        p_neg = -torch.clamp(neg_t * clip_max, min=0.0, max=clip_max)
        adv = torch.clamp(obj_img + (p_pos + p_neg), min=0.0, max=clip_max)
        l0 = cal_l0(p_pos, p_neg, l0_clip)
        if stp == 0:
            l0_init = l0
        ratio = l0 / l0_init
        if ratio <= l0_thresh:
            mw = 0
            if stp >= steps:
                break
        else:
            mw = mask_wt
        trans_adv.reset_img(adv, obj_mask)
        adv_scenes, masks, _, _, _ = paste(scene_imgs, trans_adv, batch_size)
        if color_aug is not None:
            adv_scenes = color_aug(adv_scenes)
        adv_cost = loss(model(adv_scenes) * masks, target)
        mask_cost = l0_mask_cost(pos_t, neg_t)
        total = adv_cost + mw * mask_cost
        opt.zero_grad()
        total.backward()
        opt.step()
        if record is not None:
            record.append((int(l0), float(mw), float(adv_cost.detach()), float(mask_cost.detach())))
    p_pos = torch.clamp(pos_t * clip_max, min=0.0, max=clip_max).detach()
    p_neg = -torch.clamp(neg_t * clip_max, min=0.0, max=clip_max).detach()
    p_pos[p_pos < l0_clip] = 0
    p_neg[p_neg > -l0_clip] = 0
    adv = torch.clamp(obj_img + (p_pos + p_neg), min=0.0, max=clip_max)
    trans_adv.reset_img(adv, obj_mask)
    z0 = random.sample(trans_ben.dist_range, batch_size)
    al = random.sample(trans_ben.angle_range, batch_size)
    if eval:
        z0[0], al[0] = 6.1, 0
    adv_scenes, _, full_mask, _, _ = paste(scene_imgs, trans_adv, batch_size, z0, al)
    obj_ben, _, _, _ = trans_ben.project(batch_size=batch_size, z0_sample=z0, alpha_sample=al)
    ben_scenes = tv082.resize(scene_imgs * (1 - full_mask) + obj_ben * full_mask, SCENE_SIZE)
    masks_out = tv082.resize(full_mask, SCENE_SIZE)
    if given_training:
        model.train()
    return adv_scenes, ben_scenes, masks_out, adv


def pgd_depth(model, images, eps=0.3, alpha=2 / 255, steps=40, random_start=True, targeted=False,
              start_noise=None, record=None):
    """PGD_depth.forward (pgd_depth.py:41-80) under Attack.__call__'s eval bracket."""
    given_training = model.training
    model.eval()
    images = tv082.resize(images, SCENE_SIZE).detach()
    depth_gt = model(images).detach()
    depth_target = torch.zeros_like(depth_gt)
    loss = nn.MSELoss()
    adv = images.clone().detach()
    if random_start:
        noise = torch.empty_like(adv).uniform_(-eps, eps) if start_noise is None else start_noise
        adv = torch.clamp(adv + noise, min=0, max=1).detach()
    for _ in range(steps):
        adv.requires_grad = True
        out = model(adv)
        cost = -loss(out, depth_target) if targeted else loss(out, depth_gt)
        grad = torch.autograd.grad(cost, adv, retain_graph=False, create_graph=False)[0]
        adv = adv.detach() + alpha * grad.sign()
        delta = torch.clamp(adv - images, min=-eps, max=eps)
        adv = torch.clamp(images + delta, min=0, max=1).detach()
        if record is not None:
            record.append(adv.clone())
    if given_training:
        model.train()
    return adv, images
