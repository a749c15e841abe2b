#!/usr/bin/env python3
"""Time the K10 ablation variants built by tools/wino_ablate.sh (durations only; outputs are garbage for N != 0).
    python3 tools/wino_ablate.py C K Ho Wo pad B N [N ...]"""
import ctypes as C
import sys

import torch

Cc, K, Ho, Wo, pad, B = (int(v) for v in sys.argv[1:7])
dev = torch.device("cuda")
H, W = Ho + 2 - 2 * pad, Wo + 2 - 2 * pad
x = torch.rand(B, Cc, H, W, device=dev) - 0.5
w = torch.rand(K, Cc, 3, 3, device=dev) - 0.5
y = torch.empty(B, K, Ho, Wo, device=dev)
import os
bias = (torch.rand(K, device=dev) - 0.5) if os.environ.get("BIAS") == "1" else None      # BIAS=1: with a bias vector
vp = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for n in sys.argv[7:]:
    lib = C.CDLL("var/libwino_abl%s.so" % n)
    lib.dmh_wino_weight_size.restype = C.c_int64
    lib.dmh_last_error.restype = C.c_char_p
    U = torch.empty(lib.dmh_wino_weight_size(K, Cc), device=dev)
    assert lib.dmh_wino_weight_transform(vp(w), K, Cc, 0, vp(U), st) == 0

    def run():
        rc = lib.dmh_wino_conv3x3(vp(x), vp(U), None if bias is None else vp(bias), B, Cc, K, H, W, pad, vp(y), st)
        assert rc == 0, lib.dmh_last_error()
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    bits = int(y.view(torch.int32).to(torch.int64).sum())     # equal for two builds that compute the same bits
    print("ablate %-3s C%d K%d %dx%d B%d: %.1f us   bits %d" % (n, Cc, K, Ho, Wo, B, e0.elapsed_time(e1) * 100, bits), flush=True)
