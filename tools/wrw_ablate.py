#!/usr/bin/env python3
"""Time the K18 ablation variants built by tools/wrw_ablate.sh:  python3 tools/wrw_ablate.py C K Ho Wo pad B N [N ...]"""
import ctypes as C
import sys

import torch

Cc, K, Ho, Wo, pad, B = (int(v) for v in sys.argv[1:7])
dev = torch.device("cuda")
H, W = Ho + 2 - 2 * pad, Wo + 2 - 2 * pad
x = torch.rand(B, Cc, H, W, device=dev) - 0.5
gy = torch.rand(B, K, Ho, Wo, device=dev) - 0.5
dw = torch.empty(K, Cc, 3, 3, device=dev)
vp = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for n in sys.argv[7:]:
    lib = C.CDLL("var/libwrw_abl%s.so" % n)
    lib.dmh_wino_wrw_workspace_size.restype = C.c_int64
    ws = torch.empty(lib.dmh_wino_wrw_workspace_size(B, Cc, K, H, W, pad), device=dev)

    def run():
        assert lib.dmh_wino_wrw(vp(x), vp(gy), B, Cc, K, H, W, pad, vp(ws), vp(dw), st) == 0
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    print("ablate %-3s C%d K%d %dx%d B%d: %.1f us" % (n, Cc, K, Ho, Wo, B, e0.elapsed_time(e1) * 100), flush=True)
