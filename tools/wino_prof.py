#!/usr/bin/env python3
"""One K10 shape, a few launches -- the target of rocprofv3 runs (kernel trace / PMC passes).
    python3 tools/wino_prof.py C K Ho Wo pad B reps"""
import sys

import torch

sys.path.insert(0, ".")
from depthmodelhardening_amd import _native as N  # noqa: E402

C, K, Ho, Wo, pad, B, reps = (int(v) for v in (sys.argv[1:8] + ["256", "64", "80", "256", "1", "16", "3"][len(sys.argv) - 1:]))
dev = torch.device("cuda")
lib = N.lib()
H, W = Ho + 2 - 2 * pad, Wo + 2 - 2 * pad
x = torch.rand(B, C, H, W, device=dev) - 0.5
w = torch.rand(K, C, 3, 3, device=dev) - 0.5
U = torch.empty(lib.dmh_wino_weight_size(K, C), device=dev)
N.check(lib.dmh_wino_weight_transform(N.ptr(w), K, C, 0, N.ptr(U), N.stream()))
y = torch.empty(B, K, Ho, Wo, device=dev)
for _ in range(reps):
    N.check(lib.dmh_wino_conv3x3(N.ptr(x), N.ptr(U), None, B, C, K, H, W, pad, N.ptr(y), N.stream()))
torch.cuda.synchronize()
print("done", float(y.abs().mean()))
