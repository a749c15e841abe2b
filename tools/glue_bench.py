#!/usr/bin/env python3
"""Glue-kernel micro-benchmark at the decoder's real shapes (attack batch 12): time and effective GB/s."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthmodelhardening_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = "cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
stages = [(256, 256, 10, 32), (128, 128, 20, 64), (64, 64, 40, 128), (32, 64, 80, 256), (16, 0, 160, 512)]
for C1, C2, h, w in stages:
    y = torch.randn(B, C1, h, w, device=dev, requires_grad=True)
    skip = torch.randn(B, C2, 2 * h, 2 * w, device=dev, requires_grad=True) if C2 else None
    out = ops.up_cat_pad(y, skip)
    g = torch.randn_like(out)
    fb = y.numel() * 4 + (skip.numel() * 4 if C2 else 0) + out.numel() * 4
    tf = t(lambda: ops.up_cat_pad(y, skip))
    ins = [y] + ([skip] if C2 else [])
    def bw():
        o = ops.up_cat_pad(y, skip); torch.autograd.grad(o, ins, g)
    tb = t(bw) - tf
    print("up_cat_pad C1=%3d C2=%3d %3dx%3d: fwd %.1f us (%.2f TB/s)  bwd %.1f us (%.2f TB/s)" % (C1, C2, h, w, tf * 1e3, fb / tf / 1e9, tb * 1e3, (fb + y.numel() * 4) / tb / 1e9))
    z = torch.randn(B, C1 if C1 > 16 else 16, 2 * h, 2 * w, device=dev, requires_grad=True)
    o2 = ops.elu_pad(z); g2 = torch.randn_like(o2)
    te = t(lambda: ops.elu_pad(z))
    def bw2():
        o = ops.elu_pad(z); torch.autograd.grad(o, z, g2)
    tb2 = t(bw2) - te
    nb = z.numel() * 4 + o2.numel() * 4
    print("   elu_pad  C=%3d %3dx%3d: fwd %.1f us (%.2f TB/s)  bwd %.1f us (%.2f TB/s)" % (z.shape[1], 2 * h, 2 * w, te * 1e3, nb / te / 1e9, tb2 * 1e3, (nb + z.numel() * 4) / tb2 / 1e9))
# reference: plain copy of the largest tensor
x = torch.randn(B, 96, 162, 514, device=dev); yv = torch.empty_like(x)
tc = t(lambda: yv.copy_(x))
print("torch copy %d MB: %.1f us (%.2f TB/s r+w)" % (x.numel() * 4 / 1e6, tc * 1e3, 2 * x.numel() * 4 / tc / 1e9))
