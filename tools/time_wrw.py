"""Times K20 / K21 (deterministic weight gradients of the strided block-entry convolutions and of the stem) beside ATen/MIOpen
at the train pass's shapes (batch 32).  python tools/time_wrw.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthmodelhardening_amd import _native as N


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    lib = N.lib()
    for B, C, K, H, W in [(32, 64, 128, 80, 256), (32, 128, 256, 40, 128), (32, 256, 512, 20, 64)]:
        x = torch.randn(B, C, H, W, device="cuda")
        g3 = torch.randn(B, K, H // 2, W // 2, device="cuda")
        gd = torch.randn_like(g3)
        w3 = torch.randn(K, C, 3, 3, device="cuda")
        wd = torch.randn(K, C, 1, 1, device="cuda")
        ws = torch.empty(lib.dmh_down_wrw_workspace_size(B, C, K, H, W), device="cuda")
        o3, od = torch.empty_like(w3), torch.empty_like(wd)
        t_hip = timed(lambda: N.check(lib.dmh_down_wrw(N.ptr(x), N.ptr(g3), N.ptr(gd), B, C, K, H, W, N.ptr(ws), N.ptr(o3), N.ptr(od),
                                                       N.stream())))
        t_lib = timed(lambda: (torch.ops.aten.convolution_backward(g3, x, w3, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1,
                                                                   [False, True, False]),
                               torch.ops.aten.convolution_backward(gd, x, wd, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1,
                                                                   [False, True, False])))
        fl = 20.0 * C * g3.numel()
        print("K20 %3d->%3d @%dx%d  hip %.3f ms (%.1f TFLOP/s)   ATen 3x3+1x1 %.3f ms" % (C, K, H, W, t_hip, fl / t_hip / 1e9, t_lib))
    B, H, W = 32, 320, 1024
    x = torch.rand(B, 3, H, W, device="cuda")
    g = torch.randn(B, 64, H // 2, W // 2, device="cuda")
    w = torch.randn(64, 3, 7, 7, device="cuda")
    ws = torch.empty(lib.dmh_stem_wrw_workspace_size(B, H, W), device="cuda")
    o = torch.empty_like(w)
    t_hip = timed(lambda: N.check(lib.dmh_stem_wrw(N.ptr(x), N.ptr(g), B, H, W, 0.45, 0.225, N.ptr(ws), N.ptr(o), N.stream())))
    t_lib = timed(lambda: torch.ops.aten.convolution_backward(g, (x - 0.45) / 0.225, w, None, [2, 2], [3, 3], [1, 1], False, [0, 0],
                                                              1, [False, True, False]))
    print("K21 stem @%dx%d  hip %.3f ms (%.1f TFLOP/s of 147-tap work)   normalise + ATen %.3f ms" %
          (H, W, t_hip, 2 * 147 * g.numel() / t_hip / 1e9, t_lib))


if __name__ == "__main__":
    main()
