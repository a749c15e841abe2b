"""L0 physical-object attack: two non-negative pattern tensors optimised with Adam under a tanh
sparsity penalty that an L0-ratio threshold switches on and off.

Same surface as the reference's ``torchattacks/attacks/phy_obj_atk_l0.py:16-174``.  Per iteration:
K5 l0_compose (pattern clamp/compose + thresholded L0 count, one launch) -> K3 eot_paste -> model ->
K6 masked_sq_mean + K5 l0_mask_cost -> autograd -> Adam.  The reference reads the L0 ratio on the host
every iteration (:105-111, a device sync); here the mask weight is selected ON DEVICE and the host only
looks at the ratio when it can end the loop (stp >= steps).
"""
import random
from random import sample

import numpy as np
import torch
import torch.nn.functional as F

from ... import color_jitter, ops
from ...my_utils import object_dataset_root, ori_H, ori_W, to_device_async
from ...physicalTrans import PhysicalTrans
from ...roi import RoiPlan
from ..attack import Attack


class Phy_obj_atk_l0(Attack):
    r"""
    Distance Measure : L_0
    """

    def __init__(self, model, obj_img, obj_mask, adam_lr=0.5, steps=10, mask_wt=0.1, l0_thresh=1 / 10,
                 dist_range=list(range(5, 31, 2))):
        super().__init__("PGD", model)
        self.obj_img = obj_img.clone().detach()
        self.obj_mask = obj_mask.clone().detach()
        self.steps = steps
        self.scene_size = [320, 1024]
        self.clip_max = 1
        self.learning_rate = adam_lr
        self.mask_weight_init = mask_wt
        self.mask_weight = self.mask_weight_init
        self.l0_thresh = l0_thresh
        self.l0_clip = self.clip_max / 255.
        conf = {'path': f'{object_dataset_root}/training/calib/003086.txt'}
        self.phy_trans_adv = PhysicalTrans(self.obj_img.clone(), self.obj_mask, conf, (1, 3, ori_H, ori_W),
                                           dist_range=dist_range)
        self.phy_trans_ben = PhysicalTrans(self.obj_img, self.obj_mask, conf, (1, 3, ori_H, ori_W),
                                           dist_range=dist_range)
        # ONE random colour transform per attack object, drawn here as the reference does (:41; four random.uniform + one
        # random.shuffle of the global ``random`` generator), applied by forward(..., color_jit=True)
        self.color_aug = color_jitter.get_params((0.8, 1.2), (0.8, 1.2), (0.8, 1.2), (-0.1, 0.1))
        self.use_roi = True     # evaluate the adversarial cost on windows around the object when the model offers it
        self.shard = None       # (rank, world, group): data-parallel shared-patch mode, see Phy_obj_atk.shard
        self.trace = None  # set to a list to record (l0, mask_weight, adv_cost, mask_cost) per iteration
        self.grad_trace = None  # set to a list to record the two pattern gradients Adam is handed, per iteration (tests)

    def cal_l0(self):
        """Number of pixels whose thresholded pattern is non-zero (:43-52), as a device tensor."""
        _, count = ops.l0_compose(self.obj_img, self.pattern_pos_tensor.detach(), self.pattern_neg_tensor.detach(),
                                  self.l0_clip)
        return count[0]

    def forward(self, images, batch_size, cfg_path=f'{object_dataset_root}/training/calib/003086.txt', eval=False,
                color_jit=False):
        img_B, img_C, img_H, img_W = images.size()
        if img_H != ori_H or img_W != ori_W:
            images = F.interpolate(images, size=[ori_H, ori_W], mode="bilinear", align_corners=False)
            print("image size inconsistent in l0 attack")
        images = images.detach().to(self.device)
        # data-parallel "shared patch" mode (see Phy_obj_atk.shard): this rank holds scenes rank, rank + world, ... of the
        # batch; rank 0's initial patterns and pose draws are the job's; the two pattern gradients are summed over the ranks
        mine, share, world, group, src = None, 1.0, 1, None, 0
        if self.shard is not None:
            import torch.distributed as dist
            rank, world, group = self.shard
            src = dist.get_global_rank(group, 0) if group is not None else 0
            mine = list(range(rank, batch_size, world))
            if not mine:
                raise RuntimeError("Phy_obj_atk_l0.shard: more ranks than attack scenes is not supported")
            share = len(mine) / float(batch_size)
        n_local = batch_size if mine is None else len(mine)
        if img_B != 1 and img_B != n_local:
            raise RuntimeError('Batch size doesn\'t match!')
        scene_imgs = images

        # numpy RNG on the host, exactly as the reference (:73-83)
        pats = []
        for _ in range(2):
            init_pattern = np.random.random(self.obj_img.size()) * self.clip_max
            init_pattern = np.clip(init_pattern, 0.0, self.clip_max) / self.clip_max
            t = torch.Tensor(init_pattern).to(self.device)
            if mine is not None:
                dist.broadcast(t, src=src, group=group)
            t.requires_grad = True
            pats.append(t)
        self.pattern_pos_tensor, self.pattern_neg_tensor = pats
        optimizer = torch.optim.Adam([self.pattern_pos_tensor, self.pattern_neg_tensor], lr=self.learning_rate,
                                     betas=(0.5, 0.9))

        pt = self.phy_trans_ben
        max_iter = self.steps * 2
        # All (z0, alpha) draws up front -> one H2D copy of the homographies.  The reference only consumes
        # a project() draw for iterations it actually runs, so the RNG state after each iteration's draws
        # is kept and restored if the loop ends early: later draws then match the reference draw for draw.
        draws, rng_states = [], [random.getstate()]
        for _ in range(max_iter):
            draws.append(pt.draw_samples(batch_size))
            rng_states.append(random.getstate())
        # The poses of the returned scenes (:161-163) are drawn AFTER the loop, i.e. from the RNG state that follows the draws
        # of the iterations that really ran (steps ... 2 steps of them, known only at the end).  One candidate per possible
        # count, each with the state it leaves behind: the loop's exit picks its own, and later draws continue from there --
        # in the one-process attack and, with rank 0's candidates broadcast, in the sharded one alike.
        finals, states_after = {}, {}
        for r in range(self.steps, max_iter + 1):
            random.setstate(rng_states[r])
            z0_f, al_f = sample(pt.dist_range, batch_size), sample(pt.angle_range, batch_size)
            if eval:                # the override belongs to GLOBAL scene 0 (:165-167): applied before the scenes are dealt out
                z0_f[0], al_f[0] = 6.1, 0
            finals[r], states_after[r] = (z0_f, al_f), random.getstate()
        if mine is not None:        # the job's draws are rank 0's (the final pose draws included); keep the own scenes' poses
            box = [(draws, finals)]
            dist.broadcast_object_list(box, src=src, group=group)
            draws = [([z[i] for i in mine], [a[i] for i in mine]) for z, a in box[0][0]]
            finals = {r: ([z[i] for i in mine], [a[i] for i in mine]) for r, (z, a) in box[0][1].items()}
            batch_size = n_local
        coeffs_host = np.stack([pt.coeffs_for(z0, al) for z0, al in draws], 0)
        coeffs = to_device_async(coeffs_host, self.device)
        l_pad, t_pad = pt.l_pad, pt.t_pad
        mask = self.obj_mask.to(self.device)
        # the adversarial cost reads the disparity under the object only: see Phy_obj_atk.forward
        plans = tabs = clean = None
        # (with color_jit the whole frame changes with the patch -- the contrast step blends with the pasted image's mean --
        # so neither the windows' "unchanged outside the box" nor the cached clean-frame features hold: whole-frame path)
        if (ops.ROI_ENABLED and self.use_roi and not color_jit and hasattr(self.model, "masked_sq_mean")
                and self.device.type == "cuda"):
            plans = [RoiPlan(pt.mask_boxes(z0, al, self.scene_size), *self.scene_size, depth=ops.ROI_DEPTH) for z0, al in draws]
            tabs = to_device_async(np.stack([p.table() for p in plans], 0), self.device)
            for p_, t_ in zip(plans, tabs):     # one H2D copy for all steps; each plan keeps ITS slice (RoiPlan.bind_table)
                p_.bind_table(t_)
            with torch.no_grad():       # the frames without the object: see Phy_obj_atk.forward
                clean, _ = ops.eot_paste(scene_imgs, self.obj_img, torch.zeros_like(mask), coeffs[0], l_pad, t_pad,
                                         self.scene_size)
        thresh = torch.full((), float(self.l0_thresh), device=self.device)      # fill kernels: no host sync
        w_on = torch.full((), float(self.mask_weight_init), device=self.device)
        w_off = torch.zeros((), device=self.device)
        l0_norm_init = None
        mw = w_on
        ran = 0
        for stp in range(max_iter):
            obj_img_adv, l0_norm = ops.l0_compose(self.obj_img, self.pattern_pos_tensor, self.pattern_neg_tensor,
                                                  self.l0_clip)
            if stp == 0:
                l0_norm_init = l0_norm
            below = (l0_norm.float() / l0_norm_init.float())[0] <= thresh
            if stp >= self.steps and bool(below):  # the only host read of the ratio (:106-109)
                mw = w_off
                break
            mw = torch.where(below, w_off, w_on)
            adv_scenes, adv_obj_mask = ops.eot_paste(scene_imgs, obj_img_adv, mask, coeffs[stp], l_pad, t_pad,
                                                     self.scene_size)
            if plans is not None:
                adv_cost = self.model.masked_sq_mean(adv_scenes, adv_obj_mask, plans[stp], tabs[stp], clean)
            else:
                if color_jit:       # :122-124 (off the hot path: composed from tensor operations, see color_jitter.py)
                    adv_scenes = self.color_aug(adv_scenes)
                adv_depth = self.model(adv_scenes)
                adv_cost = ops.masked_sq_mean(adv_depth, adv_obj_mask)
            mask_cost = ops.l0_mask_cost(self.pattern_pos_tensor, self.pattern_neg_tensor)
            total_cost = adv_cost + mw * mask_cost
            if mine is not None:    # this rank's part of the job's cost: the sum over the ranks below is the one-process gradient
                total_cost = adv_cost * share + mw * mask_cost * (1.0 / world)
            # same update as zero_grad(); total_cost.backward(); step() (:136-138), but only the two
            # pattern tensors get gradients: the reference's backward() also fills (and later discards)
            # weight gradients of the attacked model -- a third of the conv backward work
            g_pos, g_neg = torch.autograd.grad(total_cost, [self.pattern_pos_tensor, self.pattern_neg_tensor])
            if mine is not None:
                dist.all_reduce(g_pos, op=dist.ReduceOp.SUM, group=group)
                dist.all_reduce(g_neg, op=dist.ReduceOp.SUM, group=group)
            if self.grad_trace is not None:
                self.grad_trace.append((g_pos.detach().clone(), g_neg.detach().clone()))
            self.pattern_pos_tensor.grad, self.pattern_neg_tensor.grad = g_pos, g_neg
            optimizer.step()
            ran += 1
            if self.trace is not None:
                self.trace.append((int(l0_norm), float(mw), float(adv_cost), float(mask_cost)))
        random.setstate(states_after[ran])     # as if only the iterations that ran, and then the final poses, had drawn
        # the loop runs ``steps`` ... 2 ``steps`` iterations, by the patch's L0 ratio (:105-109): callers that time the attack
        # (bench.py) report how many it ran
        self.total_iterations = getattr(self, "total_iterations", 0) + ran
        self.total_calls = getattr(self, "total_calls", 0) + 1

        with torch.no_grad():
            obj_img_adv, _ = ops.l0_compose(self.obj_img, self.pattern_pos_tensor.detach(),
                                            self.pattern_neg_tensor.detach(), self.l0_clip, finalize=True)
        self.phy_trans_adv.reset_img(obj_img_adv, self.obj_mask)
        z0_sample, alpha_sample = finals[ran]
        cf = to_device_async(pt.coeffs_for(z0_sample, alpha_sample), self.device)
        with torch.no_grad():
            adv_scenes, obj_masks_out = ops.eot_paste(scene_imgs, obj_img_adv, mask, cf, l_pad, t_pad, self.scene_size)
            ben_scenes, _ = ops.eot_paste(scene_imgs, self.obj_img, mask, cf, l_pad, t_pad, self.scene_size)
        self.mask_weight = float(mw)
        return adv_scenes, ben_scenes, obj_masks_out, obj_img_adv
