#!/usr/bin/env python3
"""Time K11 ablation variants (var/libsmall_abl<N>.so; durations only).  python3 tools/small_ablate.py C K Ho Wo B N..."""
import ctypes as C
import sys

import torch

Cc, K, Ho, Wo, B = (int(v) for v in sys.argv[1:6])
dev = torch.device("cuda")
H, W = Ho + 2, Wo + 2
x = torch.rand(B, Cc, H, W, device=dev) - 0.5
w = torch.rand(K, Cc, 3, 3, device=dev) - 0.5
y = torch.empty(B, K, Ho, Wo, device=dev)
vp = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for n in sys.argv[6:]:
    lib = C.CDLL("var/libsmall_abl%s.so" % n)

    def run():
        assert lib.dmh_conv3x3_small(vp(x), vp(w), None, B, K, Cc, H, W, 0, 0, vp(y), st) == 0
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    print("ablate %s C%d K%d %dx%d B%d: %.1f us" % (n, Cc, K, Ho, Wo, B, e0.elapsed_time(e1) * 100), flush=True)
