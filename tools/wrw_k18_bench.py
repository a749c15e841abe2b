#!/usr/bin/env python3
"""K18 (Winograd-domain weight gradient) vs MIOpen at the train-pass shapes (batch 32):  python3 tools/wrw_k18_bench.py [B=32]
[nomiopen].  Prints a checksum of the result's bits per shape: DMH_WRW_FORM=0 (the two-phase loop) and the default (the
interleaved loop) must print the same ones."""
import sys

import torch

sys.path.insert(0, ".")
from depthmodelhardening_amd import _native as N  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
MIOPEN = "nomiopen" not in sys.argv
dev = torch.device("cuda")
lib = N.lib()
torch.backends.cudnn.benchmark = False
SHAPES = [(64, 64, 80, 256, 1, "layer1"), (128, 128, 40, 128, 1, "layer2"), (256, 256, 20, 64, 1, "layer3"),
          (512, 512, 10, 32, 1, "layer4"), (512, 256, 10, 32, 0, "upconv4_0"), (512, 256, 20, 64, 0, "upconv4_1"),
          (256, 128, 20, 64, 0, "upconv3_0"), (256, 128, 40, 128, 0, "upconv3_1"), (128, 64, 40, 128, 0, "upconv2_0"),
          (128, 64, 80, 256, 0, "upconv2_1"), (64, 32, 80, 256, 0, "upconv1_0"), (96, 32, 160, 512, 0, "upconv1_1")]


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


tot_m = tot_k = 0.0
for C, K, Ho, Wo, pad, name in SHAPES:
    H, W = Ho + 2 - 2 * pad, Wo + 2 - 2 * pad
    g = torch.Generator(device=dev).manual_seed(C + K)
    x = torch.rand(B, C, H, W, device=dev, generator=g) - 0.5
    w = torch.rand(K, C, 3, 3, device=dev, generator=g) - 0.5
    gy = torch.rand(B, K, Ho, Wo, device=dev, generator=g) - 0.5
    ws = torch.empty(lib.dmh_wino_wrw_workspace_size(B, C, K, H, W, pad), device=dev)
    dw = torch.empty_like(w)
    ref = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    N.check(lib.dmh_wino_wrw(N.ptr(x), N.ptr(gy), B, C, K, H, W, pad, N.ptr(ws), N.ptr(dw), N.stream()))
    err = float((dw - ref).abs().max() / ref.abs().max())
    bits = int(dw.view(torch.int32).to(torch.int64).sum())
    t_m = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1,
                                                             [False, True, False])) if MIOPEN else float("nan")
    t_k = timeit(lambda: N.check(lib.dmh_wino_wrw(N.ptr(x), N.ptr(gy), B, C, K, H, W, pad, N.ptr(ws), N.ptr(dw), N.stream())))
    fl = 2.0 * 9 * B * K * C * Ho * Wo
    tot_m += t_m
    tot_k += t_k
    print("%-10s C%4d K%4d %3dx%-4d | miopen (with transposes) %7.1f us (%5.1f TF/s) | K18 %7.1f us (%5.1f TF/s direct-equivalent) | "
          "rel err %.1e | bits %d" % (name, C, K, Ho, Wo, t_m, fl / t_m / 1e6, t_k, fl / t_k / 1e6, err, bits), flush=True)
print("sum: miopen %.0f us, K18 %.0f us" % (tot_m, tot_k))
