"""Device-resident, KITTI-shaped synthetic stereo data with the reference's adversarial hooks.

Stands in for ``KITTIRAWDataset``/``MonoDataset`` (MD2/datasets/mono_dataset.py:42-384) and the
``KittiLoader`` scene feed (dataLoader.py:107-257) when there is no KITTI on disk (BASELINE metric:
"synthetic KITTI-shaped frames").  The adversarial hooks keep the reference's signatures:

  set_adv_train(model2atk, obj_tensor, mask_tensor, args)   mono_dataset.py:147-175
  update_adv_obj(scene_imgs)                                mono_dataset.py:178-184

``next_batch`` is the GPU-side version of ``prep_adv_data`` (mono_dataset.py:186-265, SURVEY.md section 8f
rank 1): the adversarial object is pasted into frame 0, the benign one into the opposite stereo view and into
``color_ben``, by three K3 launches for the whole batch instead of 3 CPU perspective warps + PIL round
trips per sample inside DataLoader workers; camera side ("l"/"r") and do_flip are per-sample draws as in
``__getitem__``.  By default the freshly attacked patch is used immediately (the reference behaves that way with
num_workers=0); ``reference_stale_patch`` reproduces what its forked workers do (patch as of the epoch start,
SURVEY 3.1).  Not reproduced: the PIL 8-bit round trip and LANCZOS pyramids of ``preprocess`` (the paste resizes
bilinearly in the same pass; scales > 0 are 2^s box means) and ColorJitter.
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
        raw_left = kitti_like(self.pool_size, 3, ori_H, ori_W, self.device, self.gen)
        # right view = left rolled 8 px (at 1024 wide) plus independent texture, so photometric error is non-trivial
        shift = max(2, int(round(8 * ori_W / 1024.0)))
        raw_right = 0.9 * torch.roll(raw_left, shift, 3) + 0.1 * kitti_like(self.pool_size, 3, ori_H, ori_W, self.device, self.gen)
        # ONE pool [2 P, 3, 375, 1242]: left frames, then right frames.  The synthesis reads its frames out of it by index
        # (K3's scene_index) -- no index_select / side-pick copies of 32 full-resolution frames per batch
        self.raw = torch.cat([raw_left, raw_right], 0).contiguous()
        self.raw_left, self.raw_right = self.raw[:self.pool_size], self.raw[self.pool_size:]
        self.is_adv_train = False
        self.load_ben_color = False
        self.half_no_synthesis = False
        # KITTI split lines carry both camera sides and training flips half of the items (mono_dataset.py:287-304);
        # both are per-sample draws here.  Trainer switches them with --no_flip_sides (off = the round-1 behaviour).
        self.both_sides = True
        self.flip_augmentation = True
        self.reference_stale_patch = False
        self.make_depth_hints = False
        self.right_pyramid = False      # ("color", "s", s > 0): only --v1_multiscale reads them (MD2/trainer.py:478-483)
        self._epoch_patch = None
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

    def begin_epoch(self):
        """Epoch boundary.  With ``reference_stale_patch`` the synthesis keeps pasting the patch as it stood here until
        the next epoch: the reference's DataLoader workers are forked when ``enumerate(self.train_loader)`` creates the
        iterator (MD2/trainer.py:297) and never see the per-iteration ``update_adv_obj`` of the main process
        (SURVEY.md section 3.1)."""
        if self.is_adv_train:
            self._epoch_patch = self.obj_img_adv.clone()

    def draw_batch_geometry(self, batch_size):
        """Per-sample draws of one batch, in the order ``__getitem__`` makes them (mono_dataset.py:287-288, :297-304,
        then :190-215 inside prep_adv_data): side of the split line, do_flip, synthesis on/off (half_no_synthesis),
        (z0, alpha)."""
        geo = {"side": [], "flip": [], "synth": [], "z0": [], "alpha": []}
        for _ in range(batch_size):
            geo["flip"].append(self.rng.random() > 0.5 if self.flip_augmentation else False)
            geo["side"].append(self.rng.choice("lr") if self.both_sides else "l")
            geo["synth"].append(not self.half_no_synthesis or self.rng.random() > 0.5)
            geo["z0"].append(self.rng.choice(self.adv_trans.dist_range) if self.is_adv_train else 0.0)
            geo["alpha"].append(self.rng.choice(self.adv_trans.angle_range) if self.is_adv_train else 0)
        return geo

    def synthesize(self, raw_l, raw_r, geo, out_size, pool_index=None):
        """GPU-side ``prep_adv_data`` (mono_dataset.py:186-265) for a whole batch: frame 0 gets the adversarial object,
        the opposite stereo view and ``color_ben`` the benign one; with ``side == "r"`` frame 0 is the right image and
        the frame-0 geometry goes through ``project_w_trans(stereo_T)`` (:205-211), with ``do_flip`` the projected
        object and mask are mirrored onto the mirrored frame (:222-225).  Three K3 launches.
        Returns (color_aug_0, color_aug_s, color_ben_0, objmask_0).
        ``pool_index`` = (index of frame 0, index of the opposite view) per sample into ``self.raw`` (int32 device tensors):
        the frames are then read in place by K3 and raw_l / raw_r are not touched."""
        dev = self.device
        n = len(geo["side"])
        if pool_index is None:
            frame0, frame_s = self._pick_sides(raw_l, raw_r, geo["side"])
            i0 = i_s = None
        else:
            frame0 = frame_s = self.raw
            i0, i_s = pool_index
        K, T = self.adv_K, self.stereo_T
        far = np.array([1, 0, 1e7, 0, 1, 1e7, 0, 0], dtype=np.float32)      # object lands outside the frame: no synthesis
        # coefficient tables only for the cameras this batch uses (all-left batches never need the frame-0 table through
        # stereo_T, all-right ones never the plain one)
        any_l, any_r = any(sd == "l" for sd in geo["side"]), any(sd != "l" for sd in geo["side"])
        c0_adv = self.adv_trans.coeffs_for(geo["z0"], geo["alpha"], K=K) if any_l else None
        c0_adv_T = self.adv_trans.coeffs_for(geo["z0"], geo["alpha"], K=K, T=T) if any_r else None
        cs_ben = self.ben_trans.coeffs_for(geo["z0"], geo["alpha"], K=K) if any_r else None
        cs_ben_T = self.ben_trans.coeffs_for(geo["z0"], geo["alpha"], K=K, T=T) if any_l else None
        c0, cs = np.empty((n, 8), dtype=np.float32), np.empty((n, 8), dtype=np.float32)
        for i in range(n):
            left = geo["side"][i] == "l"
            c0[i] = (c0_adv[i] if left else c0_adv_T[i]) if geo["synth"][i] else far
            cs[i] = (cs_ben_T[i] if left else cs_ben[i]) if geo["synth"][i] else far
        c0, cs = to_device_async(c0, dev), to_device_async(cs, dev)
        flip = to_device_async([int(f) for f in geo["flip"]], dev, torch.int32) if any(geo["flip"]) else None
        patch = self._epoch_patch if (self.reference_stale_patch and self._epoch_patch is not None) else self.obj_img_adv
        lp, tp = self.adv_trans.l_pad, self.adv_trans.t_pad
        with torch.no_grad():
            aug0, _ = ops.eot_paste(frame0, patch, self.obj_mask, c0, lp, tp, out_size, flip, i0)
            aug_s, _ = ops.eot_paste(frame_s, self.obj_img_ben, self.obj_mask, cs, lp, tp, out_size, flip, i_s)
            ben0, mask0 = ops.eot_paste(frame0, self.obj_img_ben, self.obj_mask, c0, lp, tp, out_size, flip, i0)
        return aug0, aug_s, ben0, mask0

    def _pick_sides(self, raw_l, raw_r, sides):
        """(frame 0, opposite view) per sample.  With every sample on the same side (--no_flip_sides: always "l") the raw
        tensors are used as they are -- no blocking host-list upload, no two full-resolution select passes."""
        if all(sd == "l" for sd in sides):
            return raw_l, raw_r
        if all(sd != "l" for sd in sides):
            return raw_r, raw_l
        is_l = to_device_async([sd == "l" for sd in sides], self.device, torch.bool).view(len(sides), 1, 1, 1)
        return torch.where(is_l, raw_l, raw_r), torch.where(is_l, raw_r, raw_l)

    def next_batch(self, batch_size):
        dev, H, W = self.device, self.height, self.width
        picks = [self.rng.randrange(self.pool_size) for _ in range(batch_size)]
        geo = self.draw_batch_geometry(batch_size)
        inputs = {}
        if self.is_adv_train:
            # frame 0 / opposite view of every sample as indices into the pool (left frames first, then right): K3 reads them there
            P = self.pool_size
            both = to_device_async([[p + (0 if sd == "l" else P) for p, sd in zip(picks, geo["side"])],
                                    [p + (P if sd == "l" else 0) for p, sd in zip(picks, geo["side"])]], dev, torch.int32)
            aug0, aug_s, ben0, mask0 = self.synthesize(None, None, geo, (H, W), pool_index=(both[0], both[1]))
            inputs[("color_aug", 0, 0)] = aug0
            inputs[("color_ben", 0, 0)] = ben0
            if not self.half_no_synthesis:      # mono_dataset.py:248-250
                inputs[("color_objmask", 0, 0)] = mask0.expand(-1, 3, -1, -1)
                inputs[("objdepth", 0, 0)] = to_device_async(geo["z0"], dev, torch.float32).view(batch_size, 1)
            left, right = ben0, aug_s           # inputs[("color",0,-1)] = color_ben, ("color","s",-1) = color_aug("s"), :252-253
        else:
            idx = to_device_async(picks, dev, torch.int64)
            raw_l, raw_r = self.raw_left.index_select(0, idx), self.raw_right.index_select(0, idx)
            f0, fs = self._pick_sides(raw_l, raw_r, geo["side"])
            left = F.interpolate(f0, [H, W], mode="bilinear", align_corners=False)
            right = F.interpolate(fs, [H, W], mode="bilinear", align_corners=False)
            if any(geo["flip"]):
                fl = to_device_async(geo["flip"], dev, torch.bool).view(batch_size, 1, 1, 1)
                left, right = torch.where(fl, left.flip(3), left), torch.where(fl, right.flip(3), right)
            inputs[("color_aug", 0, 0)] = left
        fused = left.is_cuda and self.num_scales == 4 and H % 8 == 0 and W % 8 == 0
        for view, img in ((0, left), ("s", right)):
            # the three coarser levels in one pass (ops.avg_pyramid: bit-identical to F.avg_pool2d); the opposite view's levels
            # are read by --v1_multiscale only, so they are built when somebody asks for them
            inputs[("color", view, 0)] = img
            if view == "s" and not self.right_pyramid:
                continue
            levels = ops.avg_pyramid(img) if fused else [F.avg_pool2d(img, 2 ** s) for s in range(1, self.num_scales)]
            for s in range(1, self.num_scales):
                inputs[("color", view, s)] = levels[s - 1]
        if self.make_depth_hints:
            # stand-in for DepthHints' precomputed SGM estimates (DH/datasets/mono_dataset.py:368-388): a smooth depth field
            # with holes; inputs["depth_hint_mask"] = (hint > 0)
            d = F.avg_pool2d(torch.rand(batch_size, 1, H + 8, W + 8, device=dev, generator=self.gen), 9, 1)
            d = (d - d.amin((2, 3), keepdim=True)) / (d.amax((2, 3), keepdim=True) - d.amin((2, 3), keepdim=True) + 1e-12)
            depth = 1.0 / (0.01 + 9.99 * (0.02 + 0.28 * d))
            holes = F.avg_pool2d(torch.rand(batch_size, 1, H + 6, W + 6, device=dev, generator=self.gen), 7, 1) < 0.457
            inputs["depth_hint"] = torch.where(holes, torch.zeros_like(depth), depth).contiguous()
            inputs["depth_hint_mask"] = (inputs["depth_hint"] > 0).float()
        inputs.update(self._camera(batch_size))
        # stereo_T[0,3] = side_sign * baseline_sign * 0.1 (mono_dataset.py:367-373)
        sign = [(-1.0 if sd == "l" else 1.0) * (-1.0 if fl else 1.0) for sd, fl in zip(geo["side"], geo["flip"])]
        if any(v != -1.0 for v in sign):
            T = inputs["stereo_T"].clone()
            T[:, 0, 3] = to_device_async([0.1 * v for v in sign], dev, torch.float32)
            inputs["stereo_T"] = T
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
