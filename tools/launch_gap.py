#!/usr/bin/env python3
"""Device-side cost of a kernel boundary: N dependent launches of a ~40 us kernel with the host running ahead; GPU time
per launch minus the kernel's own duration = the gap.  ATen element-wise vs libdmh_hip kernels through ctypes, on the
default stream and on a created stream (development tool)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from depthmodelhardening_amd import _native as N  # noqa: E402

dev = torch.device("cuda")
lib = N.lib()
C, H, W = 64, 160, 512          # 21 MB in, 21.5 MB out: ~15 us at 3 TB/s
x = torch.randn(4, C, H, W, device=dev)
y = torch.empty_like(x)
g = torch.empty(4, C, H + 2, W + 2, device=dev)
n = 400


def run(name, fn, kernel_us=None):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("%-58s GPU %.2f us per launch (host enqueue %.2f us)" % (name, e0.elapsed_time(e1) * 1e3 / n, (t1 - t0) * 1e6 / n))


def suite(tag):
    st = N.stream()
    px, pg = N.ptr(x), N.ptr(g)
    run(tag + " aten mul(out=) only", lambda: torch.mul(x, 1.5, out=y))
    run(tag + " dmh_elu_pad_fwd only", lambda: lib.dmh_elu_pad_fwd(px, 4, C, H, W, 1, pg, st))

    def alt():
        torch.mul(x, 1.5, out=y)
        lib.dmh_elu_pad_fwd(px, 4, C, H, W, 1, pg, st)
    run(tag + " alternating (per pair / 2)", alt)


suite("[default stream]")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    suite("[created stream]")
