#!/usr/bin/env python3
"""K17 (32-output-channel Winograd-MFMA convolution) vs MIOpen at the decoder shapes it serves:
    python3 tools/wino32_bench.py [batch=12] [iters=20] [nomiopen]
Prints a checksum of the output's bits: DMH_K17_FORM=0 (two-phase loop) and the default (interleaved loop) must agree; also the
attack's window shapes (ragged tile regions, stream-K with a workspace)."""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from depthmodelhardening_amd import _native as N  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda")
lib = N.lib()
torch.backends.cudnn.benchmark = False
# (C, K, Ho, Wo, pad, name): forward shapes and the backward-data geometry of upconv(1,1)
SHAPES = [(96, 32, 160, 512, 0, "upconv1_1 fwd"), (32, 96, 162, 514, 2, "upconv1_1 bwd-data"), (64, 32, 80, 256, 0, "upconv1_0 fwd"),
          (96, 32, 92, 118, 0, "z11 window fwd"), (32, 96, 94, 120, 2, "z11 window bwd"), (64, 32, 48, 62, 0, "y10 window fwd"),
          (32, 64, 50, 64, 2, "y10 window bwd")]
MIOPEN = "nomiopen" not in sys.argv


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


for C, K, Ho, Wo, pad, name in SHAPES:
    H, W = Ho + 2 - 2 * pad, Wo + 2 - 2 * pad
    g = torch.Generator(device=dev).manual_seed(C + K)
    x = torch.rand(B, C, H, W, device=dev, generator=g) - 0.5
    w = (torch.rand(K, C, 3, 3, device=dev, generator=g) - 0.5) * (2.0 / (C * 9) ** 0.5)
    U = torch.empty(lib.dmh_wino32_weight_size(K, C), device=dev)
    N.check(lib.dmh_wino32_weight_transform(N.ptr(w), K, C, 0, N.ptr(U), N.stream()))
    y = torch.empty(B, K, Ho, Wo, device=dev)
    ref = F.conv2d(x, w, None, padding=pad)
    ws = torch.empty(8 << 20, device=dev)
    N.check(lib.dmh_wino32_conv3x3_ws(N.ptr(x), N.ptr(U), None, B, C, K, H, W, pad, N.ptr(y), N.ptr(ws), ws.numel(), N.stream()))
    err = float((y - ref).abs().max() / ref.abs().max())
    bits = int(y.view(torch.int32).to(torch.int64).sum())
    flops = 2.0 * B * K * C * 9 * Ho * Wo
    t_mi = timeit(lambda: F.conv2d(x, w, None, padding=pad)) if MIOPEN else float("nan")
    t_k = timeit(lambda: N.check(lib.dmh_wino32_conv3x3_ws(N.ptr(x), N.ptr(U), None, B, C, K, H, W, pad, N.ptr(y), N.ptr(ws),
                                                           ws.numel(), N.stream())))
    print("%-20s B%3d C%3d K%3d %3dx%-3d | miopen %7.1f us (%5.1f TF/s) | K17 %7.1f us (%5.1f TF/s direct-equivalent, %5.1f issued) | "
          "rel err %.1e | bits %d" % (name, B, C, K, Ho, Wo, t_mi, flops / t_mi / 1e6, t_k, flops / t_k / 1e6,
                                      flops / 2.25 / t_k / 1e6, err, bits),
          flush=True)
