#!/usr/bin/env python3
"""K10 (Winograd-MFMA 3x3 convolution) vs MIOpen at the convolution shapes of one adversarial-training step:
correctness against torch.nn.functional.conv2d and per-launch time of both, forward and backward-data.

    python tools/wino_bench.py [batch=12] [iters=10]
"""
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from depthmodelhardening_amd import _native as N  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 10
torch.backends.cudnn.benchmark = False
dev = torch.device("cuda")
lib = N.lib()

# (C, K, H_out, W_out, pad, name): encoder convs are pad 1, decoder convs run un-padded on pre-padded inputs
SHAPES = [
    (64, 64, 80, 256, 1, "enc layer1"), (128, 128, 40, 128, 1, "enc layer2"), (256, 256, 20, 64, 1, "enc layer3"),
    (512, 512, 10, 32, 1, "enc layer4"),
    (512, 256, 10, 32, 0, "dec upconv4_0"), (512, 256, 20, 64, 0, "dec upconv4_1"), (256, 128, 20, 64, 0, "dec upconv3_0"),
    (256, 128, 40, 128, 0, "dec upconv3_1"), (128, 64, 40, 128, 0, "dec upconv2_0"), (128, 64, 80, 256, 0, "dec upconv2_1"),
    (64, 32, 80, 256, 0, "dec upconv1_0"), (96, 32, 160, 512, 0, "dec upconv1_1"), (32, 16, 160, 512, 0, "dec upconv0_0"),
    (16, 16, 320, 1024, 0, "dec upconv0_1"), (16, 1, 320, 1024, 0, "dec dispconv0"), (32, 1, 160, 512, 0, "dec dispconv1"),
]


def transform(w, backward):
    K, C = w.shape[:2]
    n_out, n_in = (C, K) if backward else (K, C)
    U = torch.empty(lib.dmh_wino_weight_size(n_out, n_in), device=dev)
    N.check(lib.dmh_wino_weight_transform(N.ptr(w), K, C, int(backward), N.ptr(U), N.stream()))
    return U


import os
WS = torch.empty(8 << 20, device=dev) if os.environ.get("WINO_WS", "1") != "0" else None     # stream-K workspace (WINO_WS=0: whole items)


def wino_k10(x, U, bias, K, pad):
    Bn, C, H, W = x.shape
    y = torch.empty(Bn, K, H + 2 * pad - 2, W + 2 * pad - 2, device=dev)
    N.check(lib.dmh_wino_conv3x3_ws(N.ptr(x), N.ptr(U), N.ptr(bias), Bn, C, K, H, W, pad, N.ptr(y), N.ptr(WS),
                                    0 if WS is None else WS.numel(), N.stream()))
    return y


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(IT):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / IT * 1e3


if os.environ.get("WINO_SHAPES"):    # "C,K,Ho,Wo,pad;..." for scaling experiments
    SHAPES = [tuple(int(v) for v in t.split(",")) + ("custom",) for t in os.environ["WINO_SHAPES"].split(";")]
print("batch %d, %d iterations; times in us, TF/s = direct-convolution flops / time" % (B, IT), flush=True)
tot = {"mi_f": 0.0, "wi_f": 0.0, "mi_b": 0.0, "wi_b": 0.0}
for (C, K, Ho, Wo, pad, name) in SHAPES:
    H, W = Ho + 2 - 2 * pad, Wo + 2 - 2 * pad
    g = torch.Generator(device=dev).manual_seed(C * 7 + K)
    x = torch.rand(B, C, H, W, device=dev, generator=g) - 0.5
    w = (torch.rand(K, C, 3, 3, device=dev, generator=g) - 0.5) * (2.0 / (C * 9) ** 0.5)
    bias = torch.rand(K, device=dev, generator=g) - 0.5
    gy = torch.rand(B, K, Ho, Wo, device=dev, generator=g) - 0.5
    flops = 2.0 * B * K * C * 9 * Ho * Wo
    t0 = time.time()
    ref = F.conv2d(x, w, bias, padding=pad)
    xr = x.clone().requires_grad_(True)
    gref = torch.autograd.grad(F.conv2d(xr, w, None, padding=pad), xr, gy)[0]
    torch.cuda.synchronize()
    first = time.time() - t0
    small = (C == 16 and K <= 32) or (C == 32 and K <= 16) or K == 1
    if small:       # K11 (direct MFMA, filter in registers) instead of K10
        def wino(inp, U, bs, n_out, p, _w=w):
            Bn, _, Hi, Wi = inp.shape
            out = torch.empty(Bn, n_out, Hi + 2 * p - 2, Wi + 2 * p - 2, device=dev)
            if n_out == 1 and U is not None:      # K13: single output channel
                N.check(lib.dmh_conv3x3_head(N.ptr(inp), N.ptr(_w), N.ptr(bs), Bn, _w.shape[1], Hi, Wi, p, N.ptr(out),
                                             N.stream()))
                return out
            N.check(lib.dmh_conv3x3_small(N.ptr(inp), N.ptr(_w), N.ptr(bs), Bn, _w.shape[0], _w.shape[1], Hi, Wi, p,
                                          int(U is None), N.ptr(out), N.stream()))
            return out
        Uf, Ub = 1, None
        bwd_ok = (K == 16 and C <= 32) or (K == 32 and C <= 16) or (K <= 4 and C <= 32)
    else:
        wino = wino_k10
        Uf, Ub = transform(w, False), transform(w, True)
        bwd_ok = True
    got = wino(x, Uf, bias, K, pad)
    ggot = wino(gy, Ub, None, C, 2 - pad) if bwd_ok else gref
    ef = float((got - ref).abs().max() / ref.abs().max())
    eb = float((ggot - gref).abs().max() / gref.abs().max())
    mi_f = timeit(lambda: F.conv2d(x, w, bias, padding=pad))
    wi_f = timeit(lambda: wino(x, Uf, bias, K, pad))
    y = F.conv2d(xr, w, None, padding=pad)
    mi_b = timeit(lambda: torch.autograd.grad(y, xr, gy, retain_graph=True))
    wi_b = timeit(lambda: wino(gy, Ub, None, C, 2 - pad)) if bwd_ok else float('nan')
    tr = timeit(lambda: transform(w, False)) if not small else 0.0
    for k, v in zip(("mi_f", "wi_f", "mi_b", "wi_b"), (mi_f, wi_f, mi_b, wi_b)):
        tot[k] += v
    print("%-14s C%4d K%4d %4dx%-4d pad%d | fwd miopen %7.1f (%5.1f TF/s) wino %7.1f (%5.1f TF/s) err %.1e | "
          "bwd-data miopen %7.1f wino %7.1f err %.1e | w-transform %5.1f | first-call %.1fs"
          % (name, C, K, Ho, Wo, pad, mi_f, flops / mi_f / 1e6, wi_f, flops / wi_f / 1e6, ef, mi_b, wi_b, eb, tr, first),
          flush=True)
print("sum over shapes: fwd miopen %.0f wino %.0f | bwd-data miopen %.0f wino %.0f us" %
      (tot["mi_f"], tot["wi_f"], tot["mi_b"], tot["wi_b"]))
