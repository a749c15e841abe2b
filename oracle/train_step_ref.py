"""CPU restatement of one adversarial-training iteration (MD2/trainer.py:297-315): attack
(mono_dataset.py:178-184 -> Phy_obj_atk) -> process_batch (:335-375) -> backward -> Adam.

Used only as the checker / the timed CPU baseline (bench.py ``cpu_baseline``).  Test infrastructure only.
"""
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import attack_ref, loss_ref
from .synth import kitti_like, make_intrinsics, make_object


def make_train_inputs(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    left = kitti_like(B, 3, H, W, g)
    right = (0.9 * torch.roll(left, 8, 3) + 0.1 * kitti_like(B, 3, H, W, g)).contiguous()
    inputs = {("color", 0, s): (left if s == 0 else F.avg_pool2d(left, 2 ** s)) for s in range(4)}
    inputs[("color", "s", 0)] = right
    inputs[("color_aug", 0, 0)] = left
    inputs[("K", 0)], inputs[("inv_K", 0)] = make_intrinsics(B, H, W)
    T = torch.eye(4).repeat(B, 1, 1)
    T[:, 0, 3] = -0.1
    inputs["stereo_T"] = T
    return inputs


def train_step(encoder, decoder, optimizer, inputs, noise="randn", variant="md2", full=False):
    """process_batch + compute_losses + backward + step, on the CPU.  ``noise``: "randn" draws the tie-break noise as the
    reference does (trainer.py:644-645), None switches it off, a dict {scale: tensor} passes it in (already x 1e-5).
    Returns the total loss, or with ``full`` the whole losses dict (gradients stay on the parameters)."""
    feats = encoder(inputs[("color_aug", 0, 0)])
    outputs = decoder(feats)
    loss_ref.generate_images_pred(inputs, outputs)
    B, _, H, W = inputs[("color", 0, 0)].shape
    if isinstance(noise, str):
        noise = {s: torch.randn(B, 1, H, W) * 0.00001 for s in range(4)}
    losses, _ = loss_ref.compute_losses(inputs, outputs, noise=noise, variant=variant)
    optimizer.zero_grad()
    losses["loss"].backward()
    optimizer.step()
    return {k: v.detach() for k, v in losses.items()} if full else losses["loss"].detach()


def timed_iteration(model, B_train, Ba, atk_steps, H=320, W=1024, seed=1234, threads=None):
    """Time (attack with ``atk_steps`` PGD steps on ``Ba`` scenes) and (one train step on ``B_train`` images).
    Returns dict(attack_s, train_s, cores)."""
    if threads:
        torch.set_num_threads(threads)
    obj, mask = make_object()
    scenes = kitti_like(Ba, 3, 375, 1242, torch.Generator().manual_seed(seed))
    inputs = make_train_inputs(B_train, H, W, seed + 1)
    opt = torch.optim.Adam(list(model.parameters()), 1e-5)
    model.train()
    t0 = time.perf_counter()
    attack_ref.phy_obj_atk(model, obj, mask, scenes, Ba, eps=0.1, alpha=0.02, steps=atk_steps,
                           dist_range=attack_ref.TRAIN_DIST_RANGE)
    t1 = time.perf_counter()
    train_step(model.encoder, model.decoder, opt, inputs)
    t2 = time.perf_counter()
    return {"attack_s": t1 - t0, "train_s": t2 - t1, "cores": torch.get_num_threads()}


def timed_loss_path(B, H=320, W=1024, seed=1234):
    """generate_images_pred + compute_losses + backward alone (the like-for-like of K1+K2)."""
    inputs = make_train_inputs(B, H, W, seed)
    g = torch.Generator().manual_seed(seed + 7)
    outputs = {("disp", s): (0.02 + 0.2 * torch.rand(B, 1, H >> s, W >> s, generator=g)).requires_grad_(True)
               for s in range(4)}
    t0 = time.perf_counter()
    loss_ref.generate_images_pred(inputs, outputs)
    losses, _ = loss_ref.compute_losses(inputs, outputs, noise=None)
    losses["loss"].backward()
    return time.perf_counter() - t0
