"""Device-resident, KITTI-shaped synthetic stereo data with the reference's adversarial hooks.

Stands in for ``KITTIRAWDataset``/``MonoDataset`` (MD2/datasets/mono_dataset.py:42-384) and the
``KittiLoader`` scene feed (dataLoader.py:107-257) when there is no KITTI on disk (BASELINE metric:
"synthetic KITTI-shaped frames").  The adversarial hooks keep the reference's signatures:

  set_adv_train(model2atk, obj_tensor, mask_tensor, args)   mono_dataset.py:147-175
  update_adv_obj(scene_imgs)                                mono_dataset.py:178-184

``next_batch`` is the GPU-side version of ``prep_adv_data`` (mono_dataset.py:186-265, SURVEY.md section 8f
rank 1): the adversarial object is pasted into the left view, the benign one into the right view and into
``color_ben``, by three K3 launches for the whole batch instead of 3 CPU perspective warps + PIL round
trips per sample inside DataLoader workers.  The freshly attacked patch is used immediately (the
reference behaves that way with num_workers=0; with workers its forked copies lag by an epoch).
"""
import random

import numpy as np
import torch
import torch.nn.functional as F

from .. import ops
from ..my_utils import ori_H, ori_W, to_device_async, train_dist_range
from ..physicalTrans import PhysicalTrans
from ..torchattacks import Phy_obj_atk, Phy_obj_atk_l0


def kitti_like(n, c, h, w, device, gen):
    """5x5 box-blurred U[0,1): image-like SSIM statistics."""
    return F.avg_pool2d(torch.rand(n, c, h + 4, w + 4, device=device, generator=gen), 5, 1).contiguous()


def make_object(device, seed=7, h=260, w=300):
    """Object patch U[0,1) [1,3,260,300] and a filled-ellipse paint mask [1,1,260,300] (asset BMW.png is 300x260)."""
    g = torch.Generator().manual_seed(seed)
    patch = torch.rand(1, 3, h, w, generator=g)
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    mask = ((((xs - (w - 1) / 2) / 140.0) ** 2 + ((ys - (h - 1) / 2) / 110.0) ** 2) <= 1.0).float().view(1, 1, h, w)
    return patch.to(device).contiguous(), mask.to(device).contiguous()


class SyntheticKITTIDataset(object):
    def __init__(self, height, width, frame_idxs, num_scales, length, device, seed=1234, pool=None):
        self.height, self.width = height, width
        self.frame_idxs, self.num_scales = frame_idxs, num_scales
        self.length, self.device = length, torch.device(device)
        self.ori_H, self.ori_W = ori_H, ori_W
        self.gen = torch.Generator(device=self.device).manual_seed(seed)
        self.rng = random.Random(seed)
        self.pool_size = pool or 48
        self.raw_left = kitti_like(self.pool_size, 3, ori_H, ori_W, self.device, self.gen)
        # right view = left rolled 8 px (at 1024 wide) plus independent texture, so photometric error is non-trivial
        shift = max(2, int(round(8 * ori_W / 1024.0)))
        self.raw_right = (0.9 * torch.roll(self.raw_left, shift, 3) +
                          0.1 * kitti_like(self.pool_size, 3, ori_H, ori_W, self.device, self.gen)).contiguous()
        self.is_adv_train = False
        self.load_ben_color = False
        self.half_no_synthesis = False
        self.K = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
        stereo_T = np.eye(4, dtype=np.float32)
        stereo_T[0, 3] = -1 * 1 * 0.54   # side "l", mono_dataset.py:112-117
        self.stereo_T = stereo_T

    def __len__(self):
        return self.length

    # ------------------------------------------------------------------ adversarial hooks
    def set_adv_train(self, model2atk, obj_tensor, mask_tensor, args):
        if args['norm_type'] == "l_inf":
            self.depth_atk = Phy_obj_atk(model2atk, obj_tensor, mask_tensor, eps=args['epsilon'], alpha=args['alpha'],
                                         steps=args['step'], dist_range=train_dist_range)
        elif args['norm_type'] == "l_0":
            self.depth_atk = Phy_obj_atk_l0(model2atk, obj_tensor, mask_tensor, adam_lr=args["adam_lr"],
                                            steps=args["step"], mask_wt=args["mask_wt"], l0_thresh=args["l0_thresh"],
                                            dist_range=train_dist_range)
        else:
            raise RuntimeError("unknown norm_type %r" % (args['norm_type'],))
        self.load_ben_color = True
        self.half_no_synthesis = args['half_no_synthesis']
        self.adv_args = args
        self.is_adv_train = True
        self.obj_mask = mask_tensor
        self.obj_img_ben = obj_tensor
        self.obj_img_adv = self.obj_img_ben.clone()
        cfg = {'path': None}
        self.ben_trans = PhysicalTrans(self.obj_img_ben, self.obj_mask, cfg, (1, 3, ori_H, ori_W), dist_range=train_dist_range)
        self.adv_trans = PhysicalTrans(self.obj_img_adv, self.obj_mask, cfg, (1, 3, ori_H, ori_W), dist_range=train_dist_range)
        self.adv_K = self.K.copy()
        self.adv_K[0, :] *= ori_W
        self.adv_K[1, :] *= ori_H

    def update_adv_obj(self, scene_imgs):
        """Called once per training iteration: re-optimise the object patch against the current model."""
        _, _, _, obj_img_adv = self.depth_atk(scene_imgs, self.adv_args['batch_size'])
        self.obj_img_adv = obj_img_adv.detach()   # stays on the device (the reference moves it to the CPU workers)
        self.adv_trans.reset_img(self.obj_img_adv, self.obj_mask)

    # ------------------------------------------------------------------ batches
    def next_scenes(self, n):
        """n attack scenes [n,3,375,1242] (KittiLoader stand-in)."""
        idx = to_device_async([self.rng.randrange(self.pool_size) for _ in range(n)], self.device, torch.int64)
        return self.raw_left.index_select(0, idx)

    def next_batch(self, batch_size):
        dev, H, W = self.device, self.height, self.width
        idx = to_device_async([self.rng.randrange(self.pool_size) for _ in range(batch_size)], dev, torch.int64)
        raw_l, raw_r = self.raw_left.index_select(0, idx), self.raw_right.index_select(0, idx)
        inputs = {}
        if self.is_adv_train:
            z0 = [self.rng.choice(self.adv_trans.dist_range) for _ in range(batch_size)]
            al = [self.rng.choice(self.adv_trans.angle_range) for _ in range(batch_size)]
            c_l = to_device_async(self.adv_trans.coeffs_for(z0, al, K=self.adv_K), dev)
            c_r = to_device_async(self.ben_trans.coeffs_for(z0, al, K=self.adv_K, T=self.stereo_T), dev)
            lp, tp = self.adv_trans.l_pad, self.adv_trans.t_pad
            with torch.no_grad():
                left_adv, objmask = ops.eot_paste(raw_l, self.obj_img_adv, self.obj_mask, c_l, lp, tp, (H, W))
                right_ben, _ = ops.eot_paste(raw_r, self.obj_img_ben, self.obj_mask, c_r, lp, tp, (H, W))
                left_ben, _ = ops.eot_paste(raw_l, self.obj_img_ben, self.obj_mask, c_l, lp, tp, (H, W))
            inputs[("color_aug", 0, 0)] = left_adv
            inputs[("color_ben", 0, 0)] = left_ben
            inputs[("color_objmask", 0, 0)] = objmask.expand(-1, 3, -1, -1)
            inputs[("objdepth", 0, 0)] = to_device_async(z0, dev, torch.float32).view(batch_size, 1)
            left, right = left_ben, right_ben     # inputs[("color",0,-1)] = color_ben, mono_dataset.py:257-258
        else:
            left = F.interpolate(raw_l, [H, W], mode="bilinear", align_corners=False)
            right = F.interpolate(raw_r, [H, W], mode="bilinear", align_corners=False)
            inputs[("color_aug", 0, 0)] = left
        for s in range(self.num_scales):
            inputs[("color", 0, s)] = left if s == 0 else F.avg_pool2d(left, 2 ** s)
            inputs[("color", "s", s)] = right if s == 0 else F.avg_pool2d(right, 2 ** s)
        inputs.update(self._camera(batch_size))
        return inputs

    def _camera(self, batch_size):
        """K / inv_K per scale (mono_dataset.py:333-342) and stereo_T (:367-373): constant, built once."""
        if getattr(self, "_cam_cache", (None, None))[0] != batch_size:
            dev, H, W = self.device, self.height, self.width
            cam = {}
            for s in range(self.num_scales):
                K = self.K.copy()
                K[0, :] *= W // (2 ** s)
                K[1, :] *= H // (2 ** s)
                cam[("K", s)] = torch.from_numpy(K).to(dev).unsqueeze(0).repeat(batch_size, 1, 1).contiguous()
                cam[("inv_K", s)] = torch.from_numpy(np.linalg.pinv(K)).to(dev).unsqueeze(0).repeat(
                    batch_size, 1, 1).contiguous()
            T = torch.eye(4, device=dev).repeat(batch_size, 1, 1)
            T[:, 0, 3] = -0.1   # side "l", no flip: side_sign * baseline_sign * 0.1
            cam["stereo_T"] = T.contiguous()
            self._cam_cache = (batch_size, cam)
        return dict(self._cam_cache[1])
