#!/usr/bin/env python3
"""K15 (down-sampling block convolutions) vs MIOpen at the three encoder shapes: python3 tools/down_bench.py [batch=12]"""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from depthmodelhardening_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda")


def timeit(fn, it=10):
    fn()
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for (Ci, Co, H, W) in [(64, 128, 80, 256), (128, 256, 40, 128), (256, 512, 20, 64)]:
    x = torch.randn(B, Ci, H, W, device=dev, requires_grad=True)
    w3 = (torch.randn(Co, Ci, 3, 3, device=dev) * 0.05).requires_grad_(True)
    wd = (torch.randn(Co, Ci, 1, 1, device=dev) * 0.1).requires_grad_(True)
    g3 = torch.randn(B, Co, H // 2, W // 2, device=dev)
    gd = torch.randn_like(g3)
    with ops.frozen_weights():
        y3, yd = ops.down_convs(x, w3, wd)
        r3, rd = F.conv2d(x, w3, None, 2, 1), F.conv2d(x, wd, None, 2, 0)
        gx = torch.autograd.grad([y3, yd], x, [g3, gd], retain_graph=True)[0]
        rx = torch.autograd.grad([r3, rd], x, [g3, gd], retain_graph=True)[0]
        e = lambda a, b: float((a - b).abs().max() / b.abs().max())     # noqa: E731
        with torch.no_grad():
            t_k = timeit(lambda: ops.down_convs(x, w3, wd))
            t_m = timeit(lambda: (F.conv2d(x, w3, None, 2, 1), F.conv2d(x, wd, None, 2, 0)))
        t_kb = timeit(lambda: torch.autograd.grad([y3, yd], x, [g3, gd], retain_graph=True))
        t_mb = timeit(lambda: torch.autograd.grad([r3, rd], x, [g3, gd], retain_graph=True))
    fl = 2 * 10 * Ci * Co * B * (H // 2) * (W // 2)
    print("%3d->%3d @%dx%d B=%d: fwd K15 %.1f us (%.1f TFLOP/s) vs MIOpen %.1f us, err %.1e %.1e | bwd-data K15 %.1f us "
          "(%.1f TFLOP/s) vs MIOpen %.1f us, err %.1e" % (Ci, Co, H, W, B, t_k, fl / t_k / 1e6, t_m, e(y3, r3), e(yd, rd),
                                                          t_kb, fl / t_kb / 1e6, t_mb, e(gx, rx)))
