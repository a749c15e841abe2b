"""The colour augmentation of ``Phy_obj_atk_l0(..., color_jit=True)`` (reference torchattacks/attacks/phy_obj_atk_l0.py:41,122-124).

The reference builds ONE random transform in the attack's constructor -- ``torchvision.transforms.ColorJitter.get_params((0.8,
1.2), (0.8, 1.2), (0.8, 1.2), (-0.1, 0.1))``: with torchvision 0.8.2 (requirements.txt:93) four ``random.uniform`` draws
(brightness, contrast, saturation, hue factor) and a ``random.shuffle`` of the four operations -- and applies it to the pasted
scenes of every iteration when ``color_jit`` is set.  Nothing in the reference sets the flag (mono_dataset.py:182 calls the
attack without it), so this is interface completeness, off the hot path: the four operations are composed from torch tensor
operations (differentiable: the patch gradient flows through them), not fused into a kernel.

torchvision is absent from this image; the operations follow the published v0.8.2 tensor algorithms (``_blend`` with clamp to
[0, 1], ITU-R 601 grayscale 0.2989 / 0.587 / 0.114, per-image mean for the contrast, hue rotation in HSV).  The HSV -> RGB step is
written in the closed form  c_n = v - v s max(0, min(k_n, 4 - k_n, 1)),  k_n = (n + 6 h) mod 6,  n = 5, 3, 1,  which equals
torchvision's six-case table; oracle/tv082.py restates the table itself and tests/test_host_logic.py holds the two to each other.
"""
import random

import torch


class JitterParams(object):
    """The outcome of ColorJitter.get_params: the four factors and the order the operations are applied in."""

    def __init__(self, brightness, contrast, saturation, hue, order):
        self.factors = {"brightness": brightness, "contrast": contrast, "saturation": saturation, "hue": hue}
        self.order = list(order)

    def __call__(self, img):
        for name in self.order:
            img = _OPS[name](img, self.factors[name])
        return img

    def __repr__(self):
        return "JitterParams(%s, order=%s)" % (", ".join("%s=%.4f" % kv for kv in self.factors.items()), self.order)


def get_params(brightness=(0.8, 1.2), contrast=(0.8, 1.2), saturation=(0.8, 1.2), hue=(-0.1, 0.1)):
    """Same draws from ``random``, in the same order, as torchvision 0.8.2's ColorJitter.get_params: uniform x 4, then the
    shuffle of the four operations."""
    b = random.uniform(brightness[0], brightness[1])
    c = random.uniform(contrast[0], contrast[1])
    s = random.uniform(saturation[0], saturation[1])
    h = random.uniform(hue[0], hue[1])
    order = ["brightness", "contrast", "saturation", "hue"]
    random.shuffle(order)
    return JitterParams(b, c, s, h, order)


def _gray(img):
    return (0.2989 * img[..., 0:1, :, :] + 0.587 * img[..., 1:2, :, :] + 0.114 * img[..., 2:3, :, :])


def _blend(a, b, ratio):
    return (ratio * a + (1.0 - ratio) * b).clamp(0.0, 1.0)


def _brightness(img, f):
    return (f * img).clamp(0.0, 1.0)


def _contrast(img, f):
    return _blend(img, _gray(img).mean(dim=(-3, -2, -1), keepdim=True), f)


def _saturation(img, f):
    return _blend(img, _gray(img), f)


def _hue(img, f):
    if not (-0.5 <= f <= 0.5):
        raise ValueError('hue_factor ({}) is not in [-0.5, 0.5].'.format(f))
    r, g, b = img[..., 0, :, :], img[..., 1, :, :], img[..., 2, :, :]
    v, _ = img.max(dim=-3)
    lo, _ = img.min(dim=-3)
    cr = v - lo
    flat = cr == 0
    one = torch.ones_like(v)
    s = cr / torch.where(flat, one, v)
    d = torch.where(flat, one, cr)
    rc, gc, bc = (v - r) / d, (v - g) / d, (v - b) / d
    is_r, is_g = v == r, (v == g) & (v != r)
    h6 = torch.where(is_r, bc - gc, torch.where(is_g, 2.0 + rc - bc, 4.0 + gc - rc))
    h6 = torch.where(flat, torch.zeros_like(h6), h6)
    h = torch.fmod(h6 / 6.0 + 1.0, 1.0)
    h = torch.remainder(h + f, 1.0)
    out = []
    for n in (5.0, 3.0, 1.0):
        k = torch.remainder(n + 6.0 * h, 6.0)
        w = torch.minimum(torch.minimum(k, 4.0 - k), one).clamp(min=0.0)
        out.append((v - v * s * w).clamp(0.0, 1.0))
    return torch.stack(out, dim=-3)


_OPS = {"brightness": _brightness, "contrast": _contrast, "saturation": _saturation, "hue": _hue}
