#!/usr/bin/env python3
"""MIOpen weight-gradient convolutions of the train pass (batch 32), per layer shape: python3 tools/wrw_bench.py [batch=32]
(time per call incl. MIOpen's layout transposes, bytes of x + g, and the HBM-roof time of reading both once)."""
import sys

import torch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda")
# (name, C_in, C_out, H_in, W_in, k, stride, pad, count per train pass)
SHAPES = [("stem 7x7/2", 3, 64, 320, 1024, 7, 2, 3, 1),
          ("layer1 3x3", 64, 64, 80, 256, 3, 1, 1, 4), ("layer2.0 3x3/2", 64, 128, 80, 256, 3, 2, 1, 1),
          ("layer2.0 1x1/2", 64, 128, 80, 256, 1, 2, 0, 1), ("layer2 3x3", 128, 128, 40, 128, 3, 1, 1, 3),
          ("layer3.0 3x3/2", 128, 256, 40, 128, 3, 2, 1, 1), ("layer3.0 1x1/2", 128, 256, 40, 128, 1, 2, 0, 1),
          ("layer3 3x3", 256, 256, 20, 64, 3, 1, 1, 3), ("layer4.0 3x3/2", 256, 512, 20, 64, 3, 2, 1, 1),
          ("layer4.0 1x1/2", 256, 512, 20, 64, 1, 2, 0, 1), ("layer4 3x3", 512, 512, 10, 32, 3, 1, 1, 3),
          ("upconv4_0", 512, 256, 12, 34, 3, 1, 0, 1), ("upconv4_1", 512, 256, 22, 66, 3, 1, 0, 1),
          ("upconv3_0", 256, 128, 22, 66, 3, 1, 0, 1), ("upconv3_1", 256, 128, 42, 130, 3, 1, 0, 1),
          ("upconv2_0", 128, 64, 42, 130, 3, 1, 0, 1), ("upconv2_1", 128, 64, 82, 258, 3, 1, 0, 1),
          ("upconv1_0", 64, 32, 82, 258, 3, 1, 0, 1), ("upconv1_1", 96, 32, 162, 514, 3, 1, 0, 1),
          ("upconv0_0", 32, 16, 162, 514, 3, 1, 0, 1), ("upconv0_1", 16, 16, 322, 1026, 3, 1, 0, 1),
          ("dispconv3", 128, 1, 42, 130, 3, 1, 0, 1), ("dispconv2", 64, 1, 82, 258, 3, 1, 0, 1),
          ("dispconv1", 32, 1, 162, 514, 3, 1, 0, 1), ("dispconv0", 16, 1, 322, 1026, 3, 1, 0, 1)]


def timeit(fn, it=5):
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


tot = 0.0
for name, ci, co, h, w, k, st, pad, cnt in SHAPES:
    x = torch.randn(B, ci, h, w, device=dev)
    wt = torch.randn(co, ci, k, k, device=dev)
    ho, wo = (h + 2 * pad - k) // st + 1, (w + 2 * pad - k) // st + 1
    g = torch.randn(B, co, ho, wo, device=dev)
    t = timeit(lambda: torch.ops.aten.convolution_backward(g, x, wt, None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1,
                                                           [False, True, False]))
    nb = 4 * (x.numel() + g.numel())
    fl = 2 * ci * co * k * k * B * ho * wo
    tot += t * cnt
    print("%-16s %3d->%3d @%3dx%4d x%d: %7.1f us  (%.1f TFLOP/s; x+g %.0f MB = %.0f us at 4.5 TB/s)" % (
        name, ci, co, h, w, cnt, t, fl / t / 1e6, nb / 1e6, nb / 4.5e6))
print("total per train pass: %.2f ms" % (tot / 1e3))

# K13 weight gradient (ops.conv3x3 with one output channel) against the MIOpen rows above
sys.path.insert(0, ".")
from depthmodelhardening_amd import ops  # noqa: E402
for name, ci, co, h, w, k, st, pad, cnt in SHAPES:
    if not (co == 1 or (co == 16 and ci in (16, 32))):      # K13 heads, K16 last decoder stage
        continue
    x = torch.randn(B, ci, h, w, device=dev)
    wt = torch.randn(co, ci, 3, 3, device=dev, requires_grad=True)
    bs = torch.randn(co, device=dev, requires_grad=True)
    y = ops.conv3x3(x, wt, bs, pad)
    g = torch.randn_like(y)
    t = timeit(lambda: torch.autograd.grad(y, [wt, bs], g, retain_graph=True))
    print("%-16s K13 / K16 weight+bias gradient through autograd: %7.1f us" % (name, t))
