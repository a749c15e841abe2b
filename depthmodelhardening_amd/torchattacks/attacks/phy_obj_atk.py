"""PGD-L_inf attack on one shared object patch under expectation over physical transformations.

Same class name, constructor, call signature, return tuple and error behaviour as the reference's
``torchattacks/attacks/phy_obj_atk.py:13-123``.  The inner loop (:83-101) is re-built on the HIP kernels:

    per step   K3 eot_paste (pad + perspective + composite + resize, one launch for all B samples)
               -> model (PyTorch/MIOpen) -> K6 masked_sq_mean -> autograd (K6 bwd, model bwd, K3 bwd)
               -> K4 pgd_linf_step

instead of B x 2 ``perspective`` calls, a composite, two ``Resize`` and five element-wise kernels.
The (z0, alpha) draws use ``random.sample`` in the reference's order; the per-step homography
coefficients for the whole attack are computed on the host up front and shipped in ONE H2D copy.
"""
from random import sample

import numpy as np
import torch

from ... import ops
from ...my_utils import object_dataset_root, ori_H, ori_W, to_device_async
from ...physicalTrans import PhysicalTrans
from ...roi import RoiPlan, common_size_plans
from ..attack import Attack


class Phy_obj_atk(Attack):
    r"""
    Distance Measure : Linf

    Arguments:
        model (nn.Module): model to attack.
        obj_img (1x3xHxW), obj_mask (1x1xHxW): object patch and its paint mask.
        eps (float): maximum perturbation. (Default: 0.3)
        alpha (float): step size. (Default: 2/255)
        steps (int): number of steps. (Default: 40)
        random_start (bool): using random initialization of delta. (Default: True)
    """

    def __init__(self, model, obj_img, obj_mask, eps=0.3,
                 alpha=2 / 255, steps=40, random_start=True, dist_range=list(range(5, 31, 2))):
        super().__init__("PGD", model)
        self.obj_img = obj_img
        self.obj_mask = obj_mask
        self.eps = eps
        self.alpha = alpha
        self.steps = steps
        self.random_start = random_start
        self._supported_mode = ['default', 'targeted']
        self._targeted = True
        self.scene_size = [320, 1024]
        self.random_start_noise = None  # test hook: a tensor here replaces the uniform_(-eps, eps) draw
        self.trace = None       # test hook: set to a list to record (cost, patch gradient) of every step
        # (z0, alpha) are drawn WITHOUT replacement from 25 distances / 13 angles (physicalTrans.py:150,155), so the
        # reference raises ValueError beyond 13 scenes.  pose_group = g (<= 13) lifts that for larger batches: every run
        # of g consecutive scenes gets its own draw without replacement (physical_adv_training at batch 32: 13 + 13 + 6).
        # None = the reference's behaviour.
        self.pose_group = None
        self.use_roi = True     # evaluate the cost on windows around the object when the model offers masked_sq_mean
        # use_graph: give all steps of the attack the same window sizes (roi.common_size_plans), run step 0 eagerly, capture
        # step 1 in a HIP graph and replay it for the others -- ~130 kernel launches per step leave the host as ONE graph
        # launch (the step's Python + ctypes enqueue, ~2.5 ms, is what bounds a rank whose GPU share is small: DESIGN.md
        # section 7).  Same arithmetic as the eager loop on the same windows, bit for bit.  Off by default: at the headline
        # batch the GPU is the limiter and the common windows are a few per cent larger than each step's own.
        self.use_graph = False
        self.common_windows = False     # the common-size window plans without the graph (tests: the eager twin of use_graph)
        self._graph_pool = None
        self._model_negates = None      # does model.masked_sq_mean take negate=...?  (asked once)
        self._capture_fault = False     # test hook: make the capture of _graph_steps fail after its first launch
        self.graph_failure = None       # why use_graph switched itself off (a failed capture), else None
        self._one = None                # the constant 1 handed to autograd.grad as d cost / d cost (made once per attack)
        self._graph = None      # (graph of the previous attack, event behind its last replay): destroyed once it has run
        # Data-parallel "shared patch" mode: shard = (rank, world, process group or None).  The
        # reference attacks ONE patch on batch_size scenes per iteration (MD2/trainer.py:300-307, mono_dataset.py:178-184);
        # with a shard every rank holds scenes rank, rank + world, ... of that batch (``images`` = its own scenes), the pose
        # draws and the random start come from rank 0, and the patch gradient is summed over the ranks before the sign
        # step: all ranks end with the same patch -- the patch of the one-process attack on the concatenated scenes.
        self.shard = None
        conf = {'path': f'{object_dataset_root}/training/calib/003086.txt'}
        self.phy_trans_adv = PhysicalTrans(self.obj_img.clone(), self.obj_mask, conf, (1, 3, ori_H, ori_W),
                                           dist_range=dist_range)
        self.phy_trans_ben = PhysicalTrans(self.obj_img, self.obj_mask, conf, (1, 3, ori_H, ori_W),
                                           dist_range=dist_range)

    def _neg_cost(self, adv, m, plan, tab, clean):
        """-mean((disp * mask)^2) on the plan's windows (phy_obj_atk.py:94-95); a model whose masked_sq_mean takes ``negate``
        applies the sign inside its cost kernel (no element-wise launch for it, forward or backward)."""
        if self._model_negates is None:
            import inspect
            try:
                self._model_negates = "negate" in inspect.signature(self.model.masked_sq_mean).parameters
            except (TypeError, ValueError):
                self._model_negates = False
        if self._model_negates:
            return self.model.masked_sq_mean(adv, m, plan, tab, clean, negate=True)
        return -self.model.masked_sq_mean(adv, m, plan, tab, clean)

    def _draw(self, batch_size, explicit=False):
        """One set of (z0, alpha) for ``batch_size`` scenes in the reference's RNG order: project()'s draw
        (physicalTrans.py:146-155), or with ``explicit`` the two ``sample`` calls of phy_obj_atk.py:108-109."""
        pt, g = self.phy_trans_ben, self.pose_group
        sizes = [batch_size] if not g or batch_size <= g else [min(g, batch_size - lo) for lo in range(0, batch_size, g)]
        z0s, als = [], []
        for n in sizes:
            z0, al = (sample(pt.dist_range, n), sample(pt.angle_range, n)) if explicit else pt.draw_samples(n)
            z0s += list(z0)
            als += list(al)
        return z0s, als

    def _coeffs(self, samples):
        """One device tensor [len(samples), B, 8] for a list of (z0, alpha) sample lists."""
        host = np.stack([self.phy_trans_ben.coeffs_for(z0, al) for z0, al in samples], 0)
        return to_device_async(host, self.device)

    def forward(self, images, batch_size, cfg_path=f'{object_dataset_root}/training/calib/003086.txt', eval=False):
        r"""
        images: scene image, 1*3*375*1242 (tiled over the batch) or batch_size*3*375*1242.
        In eval mode the first object position / angle of the returned scenes is fixed (7 m, 0 deg).
        """
        images = images.detach().to(self.device)
        mine, share = None, 1.0
        if self.shard is not None:
            import torch.distributed as dist
            rank, world, group = self.shard
            mine = list(range(rank, batch_size, world))         # this rank's scenes of the global batch
            share = len(mine) / float(batch_size)               # its part of the global mean of the cost
        n_local = batch_size if mine is None else len(mine)
        if images.size()[0] != 1 and images.size()[0] != n_local:
            raise RuntimeError('Batch size doesn\'t match!')
        scene_imgs = images  # a single scene is broadcast inside the kernel (no torch.cat copy)

        obj_img_adv = self.obj_img.clone().detach()
        if self.random_start:
            noise = self.random_start_noise
            if noise is None:
                noise = torch.empty_like(obj_img_adv).uniform_(-self.eps, self.eps)
            noise = noise.to(self.device)
            if mine is not None:
                dist.broadcast(noise, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            obj_img_adv = torch.clamp(obj_img_adv + noise, min=0, max=1).detach()

        # every (z0, alpha) draw of the attack, in the reference's order: one project() per step
        # (physicalTrans.py:150,155), then the two explicit draws for the returned scenes (:108-109)
        pt = self.phy_trans_ben
        draws = [self._draw(batch_size) for _ in range(self.steps)]
        z0_sample, alpha_sample = self._draw(batch_size, explicit=True)
        if eval:
            z0_sample[0] = 7
            alpha_sample[0] = 0
        if mine is not None:        # rank 0's draws for the whole batch; every rank keeps the poses of its own scenes
            box = [draws + [(z0_sample, alpha_sample)]]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            allp = [([z[i] for i in mine], [a[i] for i in mine]) for z, a in box[0]]
            draws, (z0_sample, alpha_sample) = allp[:-1], allp[-1]
            if n_local == 0:        # more ranks than scenes: this rank only takes part in the exchange
                return self._shard_without_scenes(obj_img_adv, dist, group)
            batch_size = n_local
        coeffs = self._coeffs(draws + [(z0_sample, alpha_sample)])
        l_pad, t_pad = pt.l_pad, pt.t_pad
        mask = self.obj_mask.to(self.device)

        # the cost reads the disparity under the object only: a model that can evaluate mean((disp * mask)^2) on windows
        # around the object (DepthModelWrapper.masked_sq_mean: exact) gets the per-step boxes, all tables in one H2D copy
        plans = tabs = clean = None
        graph = False
        if ops.ROI_ENABLED and self.use_roi and hasattr(self.model, "masked_sq_mean") and self.device.type == "cuda":
            boxes = [pt.mask_boxes(z0, al, self.scene_size) for z0, al in draws]
            if self.use_graph or self.common_windows:
                plans = common_size_plans(boxes, *self.scene_size, depth=ops.ROI_DEPTH)
                # a graph holds no collective (shard), no host read (trace), and needs a step to replay
                graph = bool(self.use_graph and plans is not None and mine is None and self.trace is None and self.steps >= 3
                             and not ops.profiling_every_launch())
            if plans is None:
                plans = [RoiPlan(b, *self.scene_size, depth=ops.ROI_DEPTH) for b in boxes]
            tabs = to_device_async(np.stack([p.table() for p in plans], 0), self.device)
            if not graph:
                for p_, t_ in zip(plans, tabs):     # one H2D copy for all steps; each plan keeps ITS slice (RoiPlan.bind_table)
                    p_.bind_table(t_)
            # the frames without the object (a paste with an all-zero mask: scene (1 - 0) + patch 0, then the same Resize):
            # every step's pasted frames equal them outside the step's boxes, so the model may start from their features
            with torch.no_grad():
                clean, _ = ops.eot_paste(scene_imgs, self.obj_img, torch.zeros_like(mask), coeffs[0], l_pad, t_pad,
                                         self.scene_size)

        # d cost / d cost = 1 for every step: handed to autograd.grad as a tensor made once per attack (autograd otherwise fills
        # a fresh one-element tensor per step: one more launch in a chain of ~120 short dependent ones)
        self._one = torch.ones((), device=self.device, dtype=torch.float32)
        first = 0
        if graph:
            obj_img_adv, first = self._graph_steps(scene_imgs, obj_img_adv, mask, coeffs, plans[0], tabs, clean, l_pad, t_pad)
            if first < self.steps:      # the capture failed: the eager loop takes over on the same common-size plans
                plans[0].table_rewritten = False
                for p_, t_ in zip(plans, tabs):
                    p_.bind_table(t_)
        for s in range(first, self.steps):
            obj_img_adv.requires_grad_()
            adv_scenes, obj_masks_out = ops.eot_paste(scene_imgs, obj_img_adv, mask, coeffs[s], l_pad, t_pad,
                                                      self.scene_size)
            if plans is not None:
                cost = self._neg_cost(adv_scenes, obj_masks_out, plans[s], tabs[s], clean)
            else:
                adv_depth = self.model(adv_scenes)
                cost = -ops.masked_sq_mean(adv_depth, obj_masks_out)  # -MSE(adv_depth * mask, 0)
            if mine is not None:
                cost = cost * share     # the local mean's part of the mean over the global batch
            grad = torch.autograd.grad(cost, obj_img_adv, grad_outputs=self._one if cost.dim() == 0 and cost.dtype == torch.float32 else None,
                                       retain_graph=False, create_graph=False)[0]
            if mine is not None:        # 0.94 MB: the one exchange of the shared-patch attack, before the sign
                dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=group)
            if self.trace is not None:
                self.trace.append((float(cost), grad.detach().clone()))
            obj_img_adv = ops.pgd_linf_step(obj_img_adv, self.obj_img, grad, self.alpha, self.eps)

        self.phy_trans_adv.reset_img(obj_img_adv, self.obj_mask)
        with torch.no_grad():
            adv_scenes, obj_masks_out = ops.eot_paste(scene_imgs, obj_img_adv, mask, coeffs[-1], l_pad, t_pad,
                                                      self.scene_size)
            ben_scenes, _ = ops.eot_paste(scene_imgs, self.obj_img, mask, coeffs[-1], l_pad, t_pad, self.scene_size)
        return adv_scenes, ben_scenes, obj_masks_out, obj_img_adv

    def _graph_steps(self, scene_imgs, obj_img_adv, mask, coeffs, plan, tabs, clean, l_pad, t_pad):
        """All steps of the attack with ONE captured step.  The step reads its pose (homography coefficients, window origins)
        and its patch from fixed device buffers, so a replay after two small device copies IS the next step; every window plan
        of the attack has the sizes of ``plan`` (roi.common_size_plans).  Step 0 runs eagerly: it fills the caches of the
        frozen-weights scope (transformed filters, the clean frames' features), which must not be captured and replayed.
        Capture goes through CUDAGraph.capture_begin / capture_end on a side stream -- ``with torch.cuda.graph()`` synchronises
        the device and empties the allocator's cache on entry, once per attack here -- into a memory pool this attack object
        keeps, so that the graph of the next attack reuses the blocks of this one."""
        dev = self.device
        patch_in, patch_out = obj_img_adv.detach().clone(), torch.empty_like(obj_img_adv)
        coeff_cur, tab_cur = coeffs[0].clone(), tabs[0].clone()
        plan.bind_table(tab_cur)
        plan.table_rewritten = True     # consumers that keep origins for a later step must copy them (ops.CleanHead.mark)

        def step():
            p = patch_in.detach().requires_grad_(True)
            adv, m = ops.eot_paste(scene_imgs, p, mask, coeff_cur, l_pad, t_pad, self.scene_size)
            cost = self._neg_cost(adv, m, plan, tab_cur, clean)
            (grad,) = torch.autograd.grad(cost, p, grad_outputs=self._one)
            ops.pgd_linf_step(p, self.obj_img, grad, self.alpha, self.eps, out=patch_out)
            patch_in.copy_(patch_out)

        if self._graph is not None:         # the previous attack's graph: let its last replay finish before it is destroyed
            self._graph[1].synchronize()
            self._graph = None
        step()                                              # step 0, eager
        coeff_cur.copy_(coeffs[1])
        tab_cur.copy_(tabs[1])
        main = torch.cuda.current_stream(dev)
        if self._graph_pool is None:
            # the allocator drops a pool with its last graph: a one-kernel graph that is never destroyed keeps this one
            pool, side, keeper = torch.cuda.graph_pool_handle(), torch.cuda.Stream(device=dev), torch.cuda.CUDAGraph()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                keeper.capture_begin(pool=pool)
                try:
                    torch.zeros(8, device=dev)
                finally:
                    keeper.capture_end()
            self._graph_pool = (pool, side, keeper)
        pool, side, _ = self._graph_pool
        side.wait_stream(main)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.stream(side):
                ops._sk_workspace(dev)                      # this stream's stream-K workspace: allocated outside the capture
                # thread_local: a HIP call of ANOTHER thread (the process group's watchdog, the all-reduce still in flight on the
                # bucket's stream in the trainer's overlap mode) must not invalidate this thread's capture
                g.capture_begin(pool=pool, capture_error_mode="thread_local")
                try:
                    if self._capture_fault:                 # test hook: a capture that dies half way
                        ops.eot_paste(scene_imgs, patch_in, mask, coeff_cur, l_pad, t_pad, self.scene_size)
                        raise RuntimeError("injected capture fault")
                    step()
                except BaseException:
                    try:
                        g.capture_end()                     # ends the (invalidated) capture; its own error adds nothing
                    except Exception:
                        pass
                    raise
                g.capture_end()
        except RuntimeError as e:
            # Nothing of the captured step has executed: the device holds the state step 0 left (patch_in = the patch after
            # step 0, the encoder head's bookkeeping copies of step 0's origins).  Hand the attack back to the eager loop and
            # stop trying: a stack that cannot capture this step will not capture the next attack's either.
            main.wait_stream(side)
            self.use_graph = False
            self.graph_failure = "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")
            import warnings
            warnings.warn("Phy_obj_atk: HIP-graph capture of the attack step failed (%s); continuing with eager launches"
                          % self.graph_failure)
            return patch_in.clone(), 1
        main.wait_stream(side)
        g.replay()                                          # step 1 (capturing executes nothing)
        for s in range(2, self.steps):
            coeff_cur.copy_(coeffs[s])
            tab_cur.copy_(tabs[s])
            g.replay()
        out = patch_in.clone()
        done = torch.cuda.Event()
        done.record(main)
        self._graph = (g, done)
        return out, self.steps

    def _shard_without_scenes(self, obj_img_adv, dist, group):
        """A rank whose share of the attack batch is empty (world > batch_size): it contributes a zero gradient to every
        step's sum and follows the patch."""
        for _ in range(self.steps):
            grad = torch.zeros_like(obj_img_adv)
            dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=group)
            obj_img_adv = ops.pgd_linf_step(obj_img_adv, self.obj_img, grad, self.alpha, self.eps)
        self.phy_trans_adv.reset_img(obj_img_adv, self.obj_mask)
        empty = obj_img_adv.new_zeros((0, 3) + tuple(self.scene_size))
        return empty, empty.clone(), obj_img_adv.new_zeros((0, 1) + tuple(self.scene_size)), obj_img_adv
