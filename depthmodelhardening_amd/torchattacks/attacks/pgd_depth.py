"""Whole-image PGD-L_inf on 320x1024 frames: the reference's ``torchattacks/attacks/pgd_depth.py:7-80``
(used by physical_adv_training.py:71-81 and simple_adv_training.py:40-41) on K6 + K4.
"""
import torch
import torch.nn.functional as F

from ... import ops
from ..attack import Attack


class PGD_depth(Attack):
    """L_inf projected gradient ascent on whole frames (resized to ``scene_size`` first).

    ``eps`` bounds |adv - image| per element, ``alpha`` is the size of one sign step, ``steps`` the number of them, and
    ``random_start`` begins from the image plus uniform noise in [-eps, eps] clipped to [0, 1].  The untargeted cost is
    MSE(model(adv), model(image)); with ``atk._targeted = True`` (how the reference's callers select it,
    physical_adv_training.py:78) the cost is -MSE(model(adv), 0), i.e. the disparity is pushed to zero.
    Constructor signature and defaults: pgd_depth.py:29-36 of the reference.
    """

    def __init__(self, model, eps=0.3, alpha=2 / 255, steps=40, random_start=True):
        super().__init__("PGD", model)
        self.eps = eps
        self.alpha = alpha
        self.steps = steps
        self.random_start = random_start
        self._supported_mode = ['default', 'targeted']
        self.scene_size = [320, 1024]
        self.random_start_noise = None  # test hook

    def forward(self, images):
        # torchvision-0.8.2 Resize on a tensor = bilinear, align_corners=False, no antialias (pgd_depth.py:45)
        images = F.interpolate(images.to(self.device), size=self.scene_size, mode="bilinear",
                               align_corners=False).detach()
        depth_gt = self.model(images).detach()
        adv_images = images.clone().detach()
        if self.random_start:
            noise = self.random_start_noise
            if noise is None:
                noise = torch.empty_like(adv_images).uniform_(-self.eps, self.eps)
            adv_images = torch.clamp(adv_images + noise.to(self.device), min=0, max=1).detach()
        for _ in range(self.steps):
            adv_images.requires_grad = True
            outputs = self.model(adv_images)
            if self._targeted:
                cost = -ops.masked_sq_mean(outputs, None)            # -MSE(outputs, 0)
            else:
                cost = ops.masked_sq_mean(outputs - depth_gt, None)  # MSE(outputs, depth_gt)
            grad = torch.autograd.grad(cost, adv_images, retain_graph=False, create_graph=False)[0]
            adv_images = ops.pgd_linf_step(adv_images, images, grad, self.alpha, self.eps)
        return adv_images, images
