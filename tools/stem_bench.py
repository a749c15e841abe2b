#!/usr/bin/env python3
"""K12 (conv1 backward-data) vs MIOpen at the attack shape: python3 tools/stem_bench.py [batch=12]"""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from depthmodelhardening_amd import _native as N  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda")
lib = N.lib()
x = torch.rand(B, 3, 320, 1024, device=dev, requires_grad=True)
w = torch.rand(64, 3, 7, 7, device=dev) - 0.5
y = F.conv2d(x, w, None, 2, 3)
gy = torch.rand_like(y)
gx = torch.empty_like(x)


def timeit(fn, it=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


mi = timeit(lambda: torch.autograd.grad(y, x, gy, retain_graph=True))
k12 = timeit(lambda: N.check(lib.dmh_conv7x7s2_bwd_data(N.ptr(gy), N.ptr(w), B, 64, 3, 320, 1024, N.ptr(gx), N.stream())))
ref = torch.autograd.grad(y, x, gy, retain_graph=True)[0]
print("conv1 backward-data B=%d: miopen %.1f us, K12 %.1f us, max rel err %.1e" %
      (B, mi, k12, float((gx - ref).abs().max() / ref.abs().max())))

# K14: normalisation + forward convolution in one launch vs ATen's (x - 0.45) / 0.225 followed by MIOpen's convolution
from depthmodelhardening_amd import ops  # noqa: E402
xd = x.detach()
with torch.no_grad():
    aten = timeit(lambda: F.conv2d((xd - 0.45) / 0.225, w, None, 2, 3))
    conv_only = timeit(lambda: F.conv2d(xd, w, None, 2, 3))
    k14 = timeit(lambda: ops.stem_conv_norm(xd, w))
    err = float((ops.stem_conv_norm(xd, w) - F.conv2d((xd - 0.45) / 0.225, w, None, 2, 3)).abs().max())
fl = 2 * 147 * B * 64 * 160 * 512
print("conv1 forward B=%d: ATen normalise + MIOpen %.1f us (convolution alone %.1f us), K14 %.1f us = %.1f TFLOP/s of 157.3; "
      "max abs err %.1e" % (B, aten, conv_only, k14, fl / k14 / 1e6, err))
