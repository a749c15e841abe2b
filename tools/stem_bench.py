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
