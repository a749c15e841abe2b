#!/usr/bin/env python3
"""Run the K3 (EOT paste) launches of one attack step at the training shape (12 scenes, 375x1242 -> 320x1024) a few
times, for rocprofv3 and for HIP-event timing.  Algorithmic bytes per sample: scene 5,589,000 +
adv 3,932,160 + mask 1,310,720 (+ patch and mask 1,248,000 once)."""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthmodelhardening_amd import ops  # noqa: E402
from depthmodelhardening_amd.datasets import make_object  # noqa: E402
from depthmodelhardening_amd.my_utils import train_dist_range  # noqa: E402
from depthmodelhardening_amd.physicalTrans import PhysicalTrans  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
scenes = F.avg_pool2d(torch.rand(n, 3, 379, 1246, device=dev, generator=g), 5, 1).contiguous()
obj, mask = make_object(dev)
pt = PhysicalTrans(obj, mask, {"path": None}, (1, 3, 375, 1242), dist_range=train_dist_range)
import random
random.seed(3)
z0, al = pt.draw_samples(n)
coeffs = torch.from_numpy(pt.coeffs_for(z0, al)).to(dev)
gadv = torch.rand(n, 3, 320, 1024, device=dev, generator=g) - 0.5


def once():
    p = obj.clone().requires_grad_(True)
    adv, m = ops.eot_paste(scenes, p, mask, coeffs, pt.l_pad, pt.t_pad, (320, 1024))
    (adv * gadv).sum().backward()
    return p.grad


for _ in range(2):
    once()
torch.cuda.synchronize()
def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def fwd_only():
    with torch.no_grad():
        ops.eot_paste(scenes, obj, mask, coeffs, pt.l_pad, pt.t_pad, (320, 1024))


tf = timed(fwd_only)
tb = timed(once) - tf
nbytes = n * (5589000 + 3932160 + 1310720) + 1248000
print("paste fwd %.1f us (%.0f GB/s of %.1f MB algorithmic), fwd+bwd - fwd %.1f us (includes autograd glue and the "
      "elementwise product of the test harness)" % (tf * 1e3, nbytes / tf / 1e6, nbytes / 1e6, tb * 1e3))
