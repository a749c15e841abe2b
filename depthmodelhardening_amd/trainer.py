"""Adversarial-training Trainer: the reference's ``MD2/trainer.py`` surface on the HIP hot path.

Same method names and dict contracts as the reference:

  generate_images_pred(inputs, outputs)      MD2/trainer.py:472-523
  compute_reprojection_loss(pred, target)    MD2/trainer.py:525-537
  compute_losses(inputs, outputs) -> dict    MD2/trainer.py:539-674  (loss, loss/{s}, sup_loss, contras_loss;
                                             writes outputs["identity_selection/{s}"])
  process_batch / run_epoch / train / set_train / set_eval / save_model / load_model / save_opts

What changed underneath: compute_losses is ONE fused HIP loss (K1 + K2 + finalise) over all scales instead
of ~300 eager kernels; the attack (dataset.update_adv_obj) runs on K3/K4/K5/K6; gradients are exchanged by a
single RCCL all-reduce of a flat bucket on a side stream (ddp.GradBucket) that overlaps the next
iteration's attack unless --sync_attack asks for the reference's strict order.
"""
import json
import os
import time

import torch
import torch.nn.functional as F
import torch.optim as optim

from . import _native as N
from . import networks, ops
from .contrastive import SimSiam
from .datasets import SyntheticKITTIDataset, make_object
from .ddp import GradBucket, average_buffers, broadcast_parameters
from .depth_model import DepthModelWrapper, import_depth_model
from .layers import SSIM, get_smooth_loss
from .my_utils import ori_H, ori_W


class StepLog:
    """JSONL step log (metrics / logging; the reference prints examples/s from time.time() around a step,
    MD2/trainer.py:309-317,706-716).  mark(name) records a HIP event on the current stream at a phase boundary; the line of
    iteration i is written when iteration i + 1 ends -- its events have long completed by then, so reading them (and the
    loss) does not stall the host.  Off (no events, no file) unless a path is given."""

    def __init__(self, path, images_per_step):
        self.file = None
        self.images = images_per_step
        self.marks, self.pending, self.t0, self.host = [], None, None, None
        if path:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            self.file = open(path, "a")
        self.cuda = torch.cuda.is_available()

    def mark(self, name):
        if self.file is None:
            return
        if self.cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
        else:
            ev = time.perf_counter()
        self.marks.append((name, ev))

    def end_step(self, step, epoch, loss):
        """Close iteration ``step``: its marks and loss are parked, the previous iteration's line is written.  The loss travels
        to a pinned host slot by an asynchronous copy (float(tensor) would wait for everything enqueued so far, i.e. drain the
        queue once per iteration: measured 91.8 instead of 86.6 ms per step at the headline configuration)."""
        if self.file is None:
            return
        now = time.perf_counter()
        self._flush()
        done = None
        if torch.is_tensor(loss) and loss.is_cuda:
            if self.host is None:
                self.host = torch.empty(2, dtype=torch.float32, pin_memory=True)
            slot = self.host[step & 1:(step & 1) + 1]
            slot.copy_(loss.detach().reshape(1), non_blocking=True)
            done = torch.cuda.Event()
            done.record()
            loss = slot
        self.pending = (step, epoch, loss, done, self.marks, self.t0, now)
        # the next iteration's first interval runs from this iteration's last mark to its "start": device time between two
        # loop bodies (nothing, unless the device had to wait for the host)
        self.marks, self.t0 = self.marks[-1:], now

    def _flush(self):
        if self.pending is None:
            return
        step, epoch, loss, done, marks, t0, now = self.pending
        if done is not None:
            done.synchronize()      # recorded an iteration ago
        phases = {}
        for (_, a), (name, b) in zip(marks[:-1], marks[1:]):
            ms = a.elapsed_time(b) if self.cuda else (b - a) * 1e3
            name = "between_steps" if name == "start" else name
            phases[name] = round(phases.get(name, 0.0) + ms, 3)
        line = {"step": step, "epoch": epoch, "loss": float(loss), "phase_ms": phases}
        # images/s from the DEVICE time of the iteration (the sum of its phases: the events bracket the whole loop body).  The
        # host's own interval is reported beside it as what it is -- enqueue time: a host that runs ahead of the device reads
        # 870 images/s on an 85 ms step (profiles/r05_steps.jsonl)
        device_ms = sum(phases.values())
        if device_ms > 0:
            line["device_ms"] = round(device_ms, 3)
            line["images_per_s"] = round(self.images / (device_ms * 1e-3), 2)
        if t0 is not None:
            line["host_enqueue_ms"] = round((now - t0) * 1e3, 3)
        self.file.write(json.dumps(line) + "\n")
        self.file.flush()
        self.pending = None

    def close(self):
        if self.file is not None:
            self._flush()
            self.file.close()
            self.file = None


class LazyOutputs(dict):
    """The ``outputs`` dict of process_batch.  Entries registered with ``lazy(key, fn)`` are produced on first access:
    the fused loss keeps its per-scale selection maps as one packed byte per pixel, and the float maps the reference
    writes to outputs["identity_selection/{s}"] (MD2/trainer.py:656-658, read by its logger only) are unpacked when --
    and only when -- somebody reads them."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._lazy = {}

    def lazy(self, key, fn):
        self._lazy[key] = fn

    def __missing__(self, key):
        if key in self._lazy:
            self[key] = self._lazy.pop(key)()
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._lazy

    def get(self, key, default=None):
        return self[key] if key in self else default


class Trainer:
    def __init__(self, options, rank=0, world_size=1, device=None, host_only=False):
        """``host_only=True`` (tests of the checkpoint layout and the gradient bucket): a non-CUDA device is accepted for the
        host-side surfaces -- save_model / load_model / save_opts, the parameter bucket, the module path of the networks; the
        hot path (attack, compute_losses, train_step) still raises "no CPU path" there."""
        self.opt = options
        self.rank, self.world_size = rank, world_size
        self.log_path = os.path.join(self.opt.log_dir, self.opt.model_name)
        assert self.opt.height % 32 == 0, "'height' must be a multiple of 32"
        assert self.opt.width % 32 == 0, "'width' must be a multiple of 32"
        self.models = {}
        self.parameters_to_train = []
        self.device = device if device is not None else torch.device("cpu" if self.opt.no_cuda else "cuda")
        if torch.device(self.device).type != "cuda" and not host_only:
            # MD2/options.py --no_cuda: the reference then trains on the CPU.  This build is the HIP hot path and nothing else;
            # say so here, not inside the first attack step (the message of _native.py's operand check)
            raise RuntimeError("libdmh_hip ops need CUDA (ROCm) tensors; got device %s -- there is no CPU path (--no_cuda "
                               "is accepted for command-line compatibility only)" % (self.device,))
        if self.opt.adv_train and self.opt.norm_type not in ("l_inf", "l_0"):
            raise RuntimeError("--adv_train needs --norm_type l_inf or l_0 (MD2/options.py:94-96 has no default; the reference "
                               "fails with a NameError at MD2/trainer.py:224)")
        self.num_scales = len(self.opt.scales)
        self.num_input_frames = len(self.opt.frame_ids)
        assert self.opt.frame_ids[0] == 0, "frame_ids must start with 0"
        self.use_pose_net = not (self.opt.use_stereo and self.opt.frame_ids == [0])
        if self.opt.use_stereo:
            self.opt.frame_ids.append("s")
        if self.use_pose_net:
            raise NotImplementedError("pose networks (monocular frames -1/+1) are outside the hot-path scope; "
                                      "train with --frame_ids 0 --use_stereo as the paper's command does "
                                      "(reference README.md:87-91)")
        if self.opt.gt_depth and self.opt.adv_train and self.opt.supervised_adv and self.opt.half_no_synthesis:
            # the reference fails later, with a KeyError on inputs[("color_objmask",0,0)]: --half_no_synthesis samples carry
            # neither the object mask nor its distance (mono_dataset.py:248-250)
            raise RuntimeError("--gt_depth needs the object mask and distance of every sample: drop --half_no_synthesis")
        # --avg_reprojection (MD2/trainer.py:593,617-621,636-639) replaces the min over the source frames by their mean;
        # with the one (stereo) source frame of this trainer the two are the same number and the fused kernel serves it;
        # over several source frames compute_losses takes the composed path (_losses_composed)
        if self.opt.v1_multiscale and (self.opt.loss_variant != "md2" or self.opt.use_depth_hints):
            raise NotImplementedError("--v1_multiscale is Monodepth2's option (MD2/trainer.py:478-483,593-596)")
        if self.opt.predictive_mask:
            assert self.opt.disable_automasking, \
                "When using predictive_mask, please disable automasking with --disable_automasking"    # MD2/trainer.py:123-125
            if self.opt.v1_multiscale:
                raise NotImplementedError("--predictive_mask is served at the frame's resolution "
                                          "(MD2/trainer.py:623-635, DH/trainer.py:674-687)")
        if self.opt.use_depth_hints and (self.opt.loss_variant != "dh" or self.opt.disable_automasking):
            raise RuntimeError("--use_depth_hints is the DepthHints trainer's option: use --loss_variant dh with "
                               "auto-masking (DH/trainer.py:71-75,557-590)")

        if self.opt.fine_tune:
            m = import_depth_model((1024, 320), pre_model_path=self.opt.load_weights_folder)
            self.models["encoder"], self.models["depth"] = m.encoder, m.decoder
        else:
            self.models["encoder"] = networks.ResnetEncoder(self.opt.num_layers, self.opt.weights_init == "pretrained")
            self.models["depth"] = networks.DepthDecoder(self.models["encoder"].num_ch_enc, self.opt.scales)
        self.models["encoder"].to(self.device)
        self.models["depth"].to(self.device)
        self.parameters_to_train += list(self.models["encoder"].parameters())
        self.parameters_to_train += list(self.models["depth"].parameters())
        self.models["DepthModelWrapper"] = DepthModelWrapper(self.models["encoder"], self.models["depth"]).to(self.device)
        if self.opt.predictive_mask:
            # "the same architecture as our depth decoder", one mask per source frame (MD2/trainer.py:127-133)
            self.models["predictive_mask"] = networks.DepthDecoder(self.models["encoder"].num_ch_enc, self.opt.scales,
                                                                   num_output_channels=len(self.opt.frame_ids) - 1)
            self.models["predictive_mask"].to(self.device)
            self.parameters_to_train += list(self.models["predictive_mask"].parameters())

        if self.opt.adv_train and self.opt.supervised_adv:
            self.gt_model = import_depth_model((1024, 320)).to(self.device)   # frozen teacher, trainer.py:93-95
            for p in self.gt_model.parameters():
                p.requires_grad_(False)
            self.gt_model.eval()
        if self.opt.contrastive_learning:
            self.models["contrastive_learning"] = SimSiam().to(self.device)
            self.parameters_to_train += list(self.models["contrastive_learning"].parameters())

        if self.world_size > 1:
            broadcast_parameters(list(self.models.values()) + ([self.gt_model] if hasattr(self, "gt_model") else []))

        self.model_optimizer = optim.Adam(self.parameters_to_train, self.opt.learning_rate)
        self.model_lr_scheduler = optim.lr_scheduler.StepLR(self.model_optimizer, self.opt.scheduler_step_size, 0.1)
        # the encoder's ImageNet ``fc`` is in parameters_to_train (trainer.py:85) but never gets a gradient:
        # it is left out of the all-reduce bucket
        fc_ids = {id(p) for p in self.models["encoder"].encoder.fc.parameters()}
        self.bucket = GradBucket([p for p in self.parameters_to_train if id(p) not in fc_ids], self.world_size)

        if self.opt.load_weights_folder is not None and not self.opt.fine_tune:
            self.load_model()

        # data
        if self.opt.dataset != "synthetic":
            raise NotImplementedError("KITTI file loaders are outside the hot-path scope (SURVEY.md section 2, rows "
                                      "12-13); use --dataset synthetic")
        self.dataset = SyntheticKITTIDataset(self.opt.height, self.opt.width, self.opt.frame_ids, 4,
                                             self.opt.synthetic_len, self.device, seed=self.opt.seed + rank)
        self.dataset.both_sides = self.dataset.flip_augmentation = not self.opt.no_flip_sides
        self.dataset.reference_stale_patch = bool(self.opt.reference_stale_patch)
        self.dataset.make_depth_hints = bool(self.opt.use_depth_hints)
        self.dataset.right_pyramid = bool(self.opt.v1_multiscale)
        self.num_total_steps = len(self.dataset) // self.opt.batch_size * self.opt.num_epochs

        if self.opt.adv_train:
            obj_tensor, mask_tensor = make_object(self.device)
            common = {"batch_size": self.opt.atk_batch_size,
                      "load_ben_color": self.opt.supervised_adv or self.opt.contrastive_learning,
                      "color_aug": self.opt.contrastive_learning, "half_no_synthesis": self.opt.half_no_synthesis}
            if self.opt.norm_type == "l_inf":   # MD2/trainer.py:199-211
                args = dict(common, norm_type="l_inf", epsilon=self.opt.atk_eps, alpha=self.opt.atk_alpha,
                            step=self.opt.atk_steps, epoch=20, adv_type='object')
            else:                               # MD2/trainer.py:212-223
                args = dict(common, norm_type="l_0", step=self.opt.atk_steps, adam_lr=self.opt.atk_adam_lr,
                            mask_wt=self.opt.atk_mask_wt, l0_thresh=self.opt.atk_l0_thresh)
            self.dataset.set_adv_train(self.models["DepthModelWrapper"], obj_tensor, mask_tensor, args)
            self.adv_args = args
            if getattr(self.opt, "shared_patch", False) and self.world_size > 1:
                self.dataset.depth_atk.shard = (self.rank, self.world_size, None)
            if getattr(self.opt, "graph_attack", False):
                if self.opt.norm_type != "l_inf":
                    raise NotImplementedError("--graph_attack is the L_inf attack's option (Phy_obj_atk.use_graph)")
                self.dataset.depth_atk.use_graph = True
            self.update_adv_obj()   # trainer.py:231-233

        # The reference builds SSIM() and per-scale BackprojectDepth / Project3D modules here (MD2/trainer.py:240-254);
        # the fused loss derives the pixel grid on chip, so those ~170 MB of device buffers (B=32) are not allocated.
        # compute_reprojection_loss (stand-alone surface) builds its SSIM lazily.
        self._ssim = None
        self.tie_break_noise = None     # test hook: four [B,1,H,W] tensors (already x 1e-5) instead of the in-kernel Philox draw
        self.timings = {}
        self.val_eval_count = 10   # evaluate_attacks(..., eval_count=10), MD2/trainer.py:465
        self.step_log = StepLog(getattr(self.opt, "step_log", "") if self.rank == 0 else "",
                                self.opt.batch_size * self.world_size)
        if self.rank == 0:
            self.save_opts()

    # ------------------------------------------------------------------ mode switches
    def set_train(self):
        for m in self.models.values():
            m.train()

    def set_eval(self):
        for m in self.models.values():
            m.eval()

    # ------------------------------------------------------------------ training loop
    def train(self):
        self.epoch = 0
        self.step = 0
        self.start_time = time.time()
        try:
            for self.epoch in range(self.opt.num_epochs):
                self.run_epoch()
                if (self.epoch + 1) % self.opt.save_frequency == 0 and self.rank == 0:
                    self.save_model()
                if self.opt.max_steps and self.step >= self.opt.max_steps:
                    break
        finally:
            self.step_log.close()       # writes the last iteration's line

    def update_adv_obj(self):
        """dataset.update_adv_obj on this iteration's attack scenes (MD2/trainer.py:300-307): --atk_batch_size scenes per
        rank, or with --shared_patch this rank's share of the job's --atk_batch_size scenes."""
        n = self.adv_args["batch_size"]
        shard = getattr(self.dataset.depth_atk, "shard", None)
        if shard is not None:
            n = len(range(shard[0], n, shard[1]))
        self.dataset.update_adv_obj(self.dataset.next_scenes(n))

    def train_step(self):
        """One iteration of run_epoch's loop body (MD2/trainer.py:297-315): attack -> forward -> loss ->
        backward -> gradient all-reduce -> Adam.  Returns the losses dict."""
        overlap = self.world_size > 1 and not self.opt.sync_attack
        log = self.step_log
        log.mark("start")
        if overlap:
            # the previous iteration left its all-reduce in flight: enqueue this iteration's attack first
            # (it reads weights one optimiser step old), then apply the averaged gradients
            if self.opt.adv_train:
                self.update_adv_obj()
                log.mark("attack")
            self._apply_pending_update()
            log.mark("all_reduce_wait+adam")
        else:
            self._apply_pending_update()
            if self.opt.adv_train:
                self.update_adv_obj()
                log.mark("attack")
        inputs = self.dataset.next_batch(self.opt.batch_size)
        outputs, losses = self.process_batch(inputs)
        log.mark("forward+loss")
        with self.bucket.released():             # model_optimizer.zero_grad(); gradients land in the flat bucket afterwards
            losses["loss"].backward()
        log.mark("backward")
        self.bucket.start_all_reduce()
        self._pending = True
        if not overlap:
            self._apply_pending_update()
            log.mark("all_reduce+adam")
        log.end_step(getattr(self, "step", 0), getattr(self, "epoch", 0), losses["loss"])
        return losses

    def warm_kernels(self):
        """One local iteration without collectives or an optimiser step: the first pass through every convolution
        makes MIOpen compile its kernels into the per-user cache (minutes on a fresh box).  bench.py lets rank 0 do
        that alone before the other ranks start, so that N ranks do not build the same kernels concurrently."""
        if self.opt.adv_train:
            shard, self.dataset.depth_atk.shard = getattr(self.dataset.depth_atk, "shard", None), None   # no collectives here
            self.update_adv_obj()
            self.dataset.depth_atk.shard = shard
        inputs = self.dataset.next_batch(self.opt.batch_size)
        _, losses = self.process_batch(inputs)
        self.bucket.zero()
        losses["loss"].backward()
        self.bucket.zero()

    def _apply_pending_update(self):
        if getattr(self, "_pending", False):
            self.bucket.finish_all_reduce()
            self.model_optimizer.step()
            self._pending = False

    def run_epoch(self):
        self.set_train()
        self.dataset.begin_epoch()     # where the reference's DataLoader forks its workers (--reference_stale_patch)
        steps = len(self.dataset) // self.opt.batch_size
        for batch_idx in range(steps):
            before_op_time = time.time()
            losses = self.train_step()
            early_phase = batch_idx % self.opt.log_frequency == 0 and self.step < 2000
            late_phase = self.step % 2000 == 0
            if (early_phase or late_phase) and self.rank == 0:
                self.log_time(batch_idx, time.time() - before_op_time, losses["loss"].detach().cpu())
                if self.opt.adv_train and self.val_eval_count > 0:
                    self._apply_pending_update()
                    self.val()
            self.step += 1
            if self.opt.max_steps and self.step >= self.opt.max_steps:
                break
        self._apply_pending_update()
        self.model_lr_scheduler.step()
        if self.world_size > 1:     # BatchNorm statistics are per rank during the epoch; the checkpoint gets their mean
            average_buffers([m for n, m in self.models.items() if n != "DepthModelWrapper"])

    def val(self):
        """Validate on a single minibatch, then evaluate the model under attack (MD2/trainer.py:435-470: an L0 attack
        with 10 steps on 8 scenes, 10 batches).  The reference prints and discards the numbers; they are returned."""
        self.set_eval()
        with torch.no_grad():
            inputs = self.dataset.next_batch(self.opt.batch_size)
            self.process_batch(inputs)
        eval_args = {"norm_type": "l_0", "step": 10, "adam_lr": 0.5, "mask_wt": 0.06, "l0_thresh": 0.1,
                     "batch_size": 8}    # hard-coded in the reference, MD2/trainer.py:452-461
        from .evaluate_depth import evaluate_attacks
        errors = evaluate_attacks(self.models['DepthModelWrapper'], eval_args, eval_count=self.val_eval_count,
                                  scene_source=self.dataset.next_scenes)
        self.set_train()
        return errors

    def process_batch(self, inputs):
        """Pass a minibatch through the network and generate images and losses (MD2/trainer.py:335-375)."""
        for key, ipt in inputs.items():
            inputs[key] = ipt.to(self.device)
        features = self.models["encoder"](inputs["color_aug", 0, 0])
        outputs = LazyOutputs(self.models["depth"](features))
        outputs["middle_features_aug"] = features
        if self.opt.contrastive_learning:
            outputs["middle_features_ben"] = self.models["encoder"](inputs["color_ben", 0, 0])
        if self.opt.predictive_mask:
            outputs["predictive_mask"] = self.models["predictive_mask"](features)       # MD2/trainer.py:362-363
        self.generate_images_pred(inputs, outputs)
        losses = self.compute_losses(inputs, outputs)
        return outputs, losses

    # ------------------------------------------------------------------ the hot path
    def _frame_T(self, inputs, outputs, frame_id):
        return inputs["stereo_T"] if frame_id == "s" else outputs[("cam_T_cam", 0, frame_id)]

    def generate_images_pred(self, inputs, outputs):
        """Generate the warped (reprojected) color images for a minibatch.

        The fused loss kernel re-derives the warp on chip and never reads these tensors, so by default only
        the free aliases (``color_identity``) are written; --materialize_warps also writes
        ("depth",0,s), ("sample",f,s), ("color",f,s) through the stand-alone warp kernel (differentiable)."""
        for scale in self.opt.scales:
            for frame_id in self.opt.frame_ids[1:]:
                if self.opt.materialize_warps:
                    depth, sample, color = ops.warp_view(
                        inputs[("color", frame_id, 0)], outputs[("disp", scale)], inputs[("K", 0)],
                        inputs[("inv_K", 0)], self._frame_T(inputs, outputs, frame_id), self.opt.height,
                        self.opt.width, self.opt.min_depth, self.opt.max_depth)
                    outputs[("depth", 0, scale)] = depth
                    outputs[("sample", frame_id, scale)] = sample
                    outputs[("color", frame_id, scale)] = color
                if not self.opt.disable_automasking:
                    outputs[("color_identity", frame_id, scale)] = inputs[("color", frame_id, 0)]

    def compute_reprojection_loss(self, pred, target):
        """Reprojection loss between a batch of predicted and target images (stand-alone surface)."""
        l1_loss = torch.abs(target - pred).mean(1, True)
        if self.opt.no_ssim:
            return l1_loss
        if self._ssim is None:
            self._ssim = SSIM().to(pred.device)
        return 0.85 * self._ssim(pred, target).mean(1, True) + 0.15 * l1_loss

    def compute_losses(self, inputs, outputs):
        """Compute the reprojection and smoothness losses for a minibatch."""
        losses = {}
        total_loss = 0
        if self.opt.adv_train and self.opt.supervised_adv:
            disp = outputs[("disp", 0)]
            with torch.no_grad():
                disp_gt = self.gt_model(inputs[("color_ben", 0, 0)])
            if self.opt.gt_depth:       # MD2/trainer.py:551-557: metric depths, the object's known distance under its mask
                loss_sup = ops.gt_depth_mse(disp, disp_gt, inputs[("color_objmask", 0, 0)], inputs[("objdepth", 0, 0)],
                                            self.opt.min_depth, self.opt.max_depth)
            else:
                loss_sup = ops.masked_sq_mean(disp_gt - disp, None)     # MSELoss(disp_gt, disp), trainer.py:559
            losses["sup_loss"] = loss_sup
            total_loss = total_loss + loss_sup
        if self.opt.adv_train and self.opt.contrastive_learning:
            wt = 1 if self.opt.loss_variant == "md2" else 0.1       # MD2/trainer.py:571 vs DH/trainer.py:617
            contras_loss = self.models['contrastive_learning'](outputs["middle_features_aug"],
                                                               outputs["middle_features_ben"]) * wt
            losses["contras_loss"] = contras_loss
            total_loss = total_loss + contras_loss
        if self.opt.adv_train and self.opt.no_original_train:
            losses["loss"] = total_loss
            return losses

        frames = self.opt.frame_ids[1:]
        if self.opt.v1_multiscale:
            return self._losses_v1_multiscale(inputs, outputs, losses, total_loss, frames)
        if self.opt.predictive_mask or (self.opt.avg_reprojection and len(frames) > 1):
            if self.opt.loss_variant != "md2":      # DepthHints forms its masks differently (DH/trainer.py:667-712)
                return self._losses_composed_dh(inputs, outputs, losses, total_loss, frames)
            return self._losses_composed(inputs, outputs, losses, total_loss, frames)
        out = ops.photometric_smooth_loss(
            inputs[("color", 0, 0)], [inputs[("color", f, 0)] for f in frames],
            [self._frame_T(inputs, outputs, f) for f in frames], inputs[("K", 0)], inputs[("inv_K", 0)],
            [outputs[("disp", s)] for s in self.opt.scales], [inputs[("color", 0, s)] for s in self.opt.scales],
            min_depth=self.opt.min_depth, max_depth=self.opt.max_depth, variant=self.opt.loss_variant,
            automask=not self.opt.disable_automasking, no_ssim=self.opt.no_ssim,
            smooth_wt=self.opt.disparity_smoothness, noise="philox" if self.tie_break_noise is None else self.tie_break_noise,
            depth_hint=inputs["depth_hint"] if self.opt.use_depth_hints else None,
            depth_hint_mask=inputs["depth_hint_mask"] if self.opt.use_depth_hints else None)
        for i, scale in enumerate(self.opt.scales):
            losses["loss/{}".format(scale)] = out.fin[N.FIN_LOSS_S + i]
            if self.opt.loss_variant == "dh":
                losses["reproj_loss/{}".format(scale)] = out.fin[N.FIN_REPROJ_S + i]
            if self.opt.use_depth_hints:
                losses["depth_hint_loss/{}".format(scale)] = out.fin[N.FIN_HINT_S + i]      # DH/trainer.py:725
                hkey = "depth_hint_pixels/{}".format(scale)

                def hint_pixels(i=i):
                    return (out.sel[i] == 3).float().unsqueeze(1)
                if isinstance(outputs, LazyOutputs):
                    outputs.lazy(hkey, hint_pixels)
                else:
                    outputs[hkey] = hint_pixels()
            if not self.opt.disable_automasking:
                def selection(i=i, multi=len(frames) > 1 or self.opt.loss_variant == "dh",
                              dh=self.opt.loss_variant == "dh"):
                    sel = out.sel[i]
                    if multi:
                        sel = (sel > 0).float()
                        if dh:
                            sel = 1 - sel                             # DH/trainer.py:703
                    return sel
                key = "identity_selection/{}".format(scale)
                if isinstance(outputs, LazyOutputs):
                    outputs.lazy(key, selection)
                else:
                    outputs[key] = selection()
        total_loss = total_loss + out.fin[N.FIN_LOSS]
        losses["loss"] = total_loss
        return losses

    def _losses_composed(self, inputs, outputs, losses, total_loss, frames):
        """The option branches of the per-scale body that the fused kernel does not carry (MD2/trainer.py:589-668):
        --predictive_mask (:623-635, with --disable_automasking) and --avg_reprojection over several source frames (:617-621,
        :637-640).  Neither is on the paper's command line; they are composed from the stand-alone operators -- the warp
        kernel (``ops.warp_view``, differentiable in the disparity), ``torch.ops.dmh.ssim_map`` behind
        ``compute_reprojection_loss``, ``torch.ops.dmh.smooth_loss`` behind ``get_smooth_loss`` -- and element-wise torch in
        the reference's order; per-pixel maps are materialised, as in the reference."""
        target = inputs[("color", 0, 0)]
        H, W = target.shape[-2:]
        automask = not self.opt.disable_automasking
        per_scale = []
        for scale in self.opt.scales:
            loss = 0
            disp = outputs[("disp", scale)]
            reproj = []
            for f in frames:
                key = ("color", f, scale)
                pred = outputs[key] if key in outputs else ops.warp_view(
                    inputs[("color", f, 0)], disp, inputs[("K", 0)], inputs[("inv_K", 0)], self._frame_T(inputs, outputs, f),
                    H, W, self.opt.min_depth, self.opt.max_depth)[2]
                reproj.append(self.compute_reprojection_loss(pred, target))
            reproj = torch.cat(reproj, 1)
            if automask:
                ident = torch.cat([self.compute_reprojection_loss(inputs[("color", f, 0)], target) for f in frames], 1)
                if self.opt.avg_reprojection:
                    ident = ident.mean(1, keepdim=True)
            elif self.opt.predictive_mask:
                mask = F.interpolate(outputs["predictive_mask"][("disp", scale)], [H, W], mode="bilinear", align_corners=False)
                reproj = reproj * mask
                loss = loss + 0.2 * F.binary_cross_entropy(mask, torch.ones_like(mask))     # pushes the mask to 1
            if self.opt.avg_reprojection:
                reproj = reproj.mean(1, keepdim=True)
            if automask:
                ident = ident + torch.randn(ident.shape, device=ident.device) * 0.00001     # breaks ties
                combined = torch.cat((ident, reproj), dim=1)
            else:
                combined = reproj
            if combined.shape[1] == 1:
                to_optimise = combined
            else:
                to_optimise, idxs = torch.min(combined, dim=1)
            if automask:
                outputs["identity_selection/{}".format(scale)] = (idxs > ident.shape[1] - 1).float()
            loss = loss + to_optimise.mean()
            norm_disp = disp / (disp.mean(2, True).mean(3, True) + 1e-7)
            loss = loss + self.opt.disparity_smoothness * get_smooth_loss(norm_disp, inputs[("color", 0, scale)]) / (2 ** scale)
            losses["loss/{}".format(scale)] = loss
            per_scale.append(loss)
        total_loss = total_loss + sum(per_scale) / self.num_scales
        losses["loss"] = total_loss
        return losses

    def _losses_composed_dh(self, inputs, outputs, losses, total_loss, frames):
        """The same option branches in DepthHints' per-scale body (DH/trainer.py:638-741), composed like _losses_composed.  What
        differs from Monodepth2's: the candidates are reduced over the source frames first (the minimum "as we go", :668-671 /
        :693-697, or the mean with --avg_reprojection), the tie-break noise has one channel, compute_loss_masks (:559-590) turns the
        argmin over [reprojection, identity, hint reprojection] into masks -- without auto-masking every pixel counts -- and the
        photometric term of a scale is sum(loss * mask) / (sum(mask) + 1e-7); --predictive_mask multiplies the per-frame losses
        before the reduction and adds 0.2 * BCE(mask, 1); --use_depth_hints adds the hint candidate (warped once, with the stereo
        pose, :510-525) and the proxy log-L1 term where it wins (:716-727)."""
        target = inputs[("color", 0, 0)]
        B, _, H, W = target.shape
        automask = not self.opt.disable_automasking
        hint_reproj = None
        if self.opt.use_depth_hints:
            if not automask:    # the reference's compute_loss_masks evaluates `if <tensor>:` there and raises (:568)
                raise RuntimeError("--use_depth_hints with --disable_automasking: DepthHints' compute_loss_masks cannot form its "
                                   "masks (depth-hints/trainer.py:568)")
            if "s" not in frames:
                raise KeyError(("color_depth_hint", "s", 0))        # the hint is only warped for the stereo frame (:513)
            from .layers import BackprojectDepth, Project3D
            cam = BackprojectDepth(B, H, W).to(target.device)(inputs["depth_hint"], inputs[("inv_K", 0)])
            grid = Project3D(B, H, W).to(target.device)(cam, inputs[("K", 0)], inputs["stereo_T"])
            pred = F.grid_sample(inputs[("color", "s", 0)], grid, padding_mode="border", align_corners=False)
            outputs[("color_depth_hint", "s", 0)] = pred
            hint_reproj = self.compute_reprojection_loss(pred, target) + 1000 * (1 - inputs["depth_hint_mask"])
        per_scale = []
        for scale in self.opt.scales:
            loss = 0
            disp = outputs[("disp", scale)]
            reproj = []
            for f in frames:
                key = ("color", f, scale)
                if key not in outputs or ("depth", 0, scale) not in outputs:
                    depth, _, pred = ops.warp_view(inputs[("color", f, 0)], disp, inputs[("K", 0)], inputs[("inv_K", 0)],
                                                   self._frame_T(inputs, outputs, f), H, W, self.opt.min_depth, self.opt.max_depth)
                    outputs.setdefault(("depth", 0, scale), depth)
                pred = outputs[key] if key in outputs else pred
                reproj.append(self.compute_reprojection_loss(pred, target))
            reproj = torch.cat(reproj, 1)
            ident = None
            if automask:
                ident = torch.cat([self.compute_reprojection_loss(inputs[("color", f, 0)], target) for f in frames], 1)
                ident = ident.mean(1, keepdim=True) if self.opt.avg_reprojection else torch.min(ident, dim=1, keepdim=True)[0]
            elif self.opt.predictive_mask:
                mask = F.interpolate(outputs["predictive_mask"][("disp", scale)], [H, W], mode="bilinear", align_corners=False)
                reproj = reproj * mask
                loss = loss + 0.2 * F.binary_cross_entropy(mask, torch.ones_like(mask))     # pushes the mask to 1
            reproj = reproj.mean(1, keepdim=True) if self.opt.avg_reprojection else torch.min(reproj, dim=1, keepdim=True)[0]
            if automask:
                ident = ident + torch.randn(ident.shape, device=ident.device) * 0.00001     # breaks ties
                cands = [reproj, ident] + ([hint_reproj] if hint_reproj is not None else [])
                idxs = torch.argmin(torch.cat(cands, dim=1), dim=1, keepdim=True)
                rmask = (idxs != 1).float()         # the auto-mask is candidate 1
            else:
                rmask = torch.ones_like(reproj)
            reproj_loss = (reproj * rmask).sum() / (rmask.sum() + 1e-7)
            outputs["identity_selection/{}".format(scale)] = (1 - rmask).float()
            losses["reproj_loss/{}".format(scale)] = reproj_loss
            loss = loss + reproj_loss
            if hint_reproj is not None:
                hmask = (idxs == 2).float()
                hint_loss = torch.log(torch.abs(inputs["depth_hint"] - outputs[("depth", 0, scale)]) + 1) \
                    * inputs["depth_hint_mask"] * hmask
                hint_loss = hint_loss.sum() / (hmask.sum() + 1e-7)
                outputs["depth_hint_pixels/{}".format(scale)] = hmask
                losses["depth_hint_loss/{}".format(scale)] = hint_loss
                loss = loss + hint_loss
            norm_disp = disp / (disp.mean(2, True).mean(3, True) + 1e-7)
            loss = loss + self.opt.disparity_smoothness * get_smooth_loss(norm_disp, inputs[("color", 0, scale)]) / (2 ** scale)
            losses["loss/{}".format(scale)] = loss
            per_scale.append(loss)
        total_loss = total_loss + sum(per_scale) / self.num_scales
        losses["loss"] = total_loss
        return losses

    def _losses_v1_multiscale(self, inputs, outputs, losses, total_loss, frames):
        """--v1_multiscale (MD2/trainer.py:478-483,593-596: Monodepth v1's multi-scale loss): scale s is a problem of its
        own at the resolution of scale s -- the source view ("color", f, s) is warped with disp_s as it is (no up-sampling)
        through K / inv_K of that scale, against the target ("color", 0, s); loss = mean_s(reproj_s + wt * smooth_s / 2^s).
        One fused K1 + K2 call per scale (single-scale launches: the disparity has the image's resolution)."""
        per_scale = []
        for i, scale in enumerate(self.opt.scales):
            out = ops.photometric_smooth_loss(
                inputs[("color", 0, scale)], [inputs[("color", f, scale)] for f in frames],
                [self._frame_T(inputs, outputs, f) for f in frames], inputs[("K", scale)], inputs[("inv_K", scale)],
                [outputs[("disp", scale)]], [inputs[("color", 0, scale)]], min_depth=self.opt.min_depth,
                max_depth=self.opt.max_depth, variant="md2", automask=not self.opt.disable_automasking,
                no_ssim=self.opt.no_ssim, smooth_wt=self.opt.disparity_smoothness / (2 ** scale), noise="philox")
            losses["loss/{}".format(scale)] = out.fin[N.FIN_LOSS_S]
            per_scale.append(out.fin[N.FIN_LOSS_S])
            if not self.opt.disable_automasking:
                sel = out.sel
                key = "identity_selection/{}".format(scale)
                if isinstance(outputs, LazyOutputs):
                    outputs.lazy(key, lambda sel=sel: sel[0] if len(frames) == 1 else (sel[0] > 0).float())
                else:
                    outputs[key] = sel[0] if len(frames) == 1 else (sel[0] > 0).float()
        total_loss = total_loss + sum(per_scale) / self.num_scales
        losses["loss"] = total_loss
        return losses

    # ------------------------------------------------------------------ logging / checkpoints
    def log_time(self, batch_idx, duration, loss):
        samples_per_sec = self.opt.batch_size * self.world_size / duration
        time_sofar = time.time() - self.start_time if hasattr(self, "start_time") else 0.0
        print("epoch {:>3} | batch {:>6} | examples/s: {:5.1f} | loss: {:.5f} | time elapsed: {:.0f}s".format(
            getattr(self, "epoch", 0), batch_idx, samples_per_sec, float(loss), time_sofar), flush=True)

    def save_opts(self):
        models_dir = os.path.join(self.log_path, "models")
        os.makedirs(models_dir, exist_ok=True)
        with open(os.path.join(models_dir, 'opt.json'), 'w') as f:
            json.dump({k: v for k, v in self.opt.__dict__.items()}, f, indent=2, default=str)

    def save_model(self):
        """weights_{epoch}/{model}.pth state_dicts (+ height/width/use_stereo in the encoder) and adam.pth --
        the reference's on-disk format (MD2/trainer.py:765-785), written by rank 0 only."""
        save_folder = os.path.join(self.log_path, "models", "weights_{}".format(self.epoch))
        os.makedirs(save_folder, exist_ok=True)
        for model_name, model in self.models.items():
            if model_name == 'DepthModelWrapper':       # a view of encoder + depth, not a model of its own (:773-774)
                continue
            to_save = model.state_dict()
            if model_name == 'encoder':
                to_save['height'] = self.opt.height
                to_save['width'] = self.opt.width
                to_save['use_stereo'] = self.opt.use_stereo
            torch.save(to_save, os.path.join(save_folder, "{}.pth".format(model_name)))
        torch.save(self.model_optimizer.state_dict(), os.path.join(save_folder, "adam.pth"))

    def load_model(self):
        """MD2/trainer.py:787-812: load the listed models (filtering unknown keys) and the Adam state."""
        folder = os.path.expanduser(self.opt.load_weights_folder)
        assert os.path.isdir(folder), "Cannot find folder {}".format(folder)
        for n in self.opt.models_to_load:
            path = os.path.join(folder, "{}.pth".format(n))
            if n not in self.models or not os.path.isfile(path):
                continue
            model_dict = self.models[n].state_dict()
            pretrained = torch.load(path, map_location="cpu")
            model_dict.update({k: v for k, v in pretrained.items() if k in model_dict})
            self.models[n].load_state_dict(model_dict)
        adam = os.path.join(folder, "adam.pth")
        if os.path.isfile(adam):
            self.model_optimizer.load_state_dict(torch.load(adam, map_location="cpu"))
