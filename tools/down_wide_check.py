#!/usr/bin/env python3
"""K15 with 16-byte loaders and the filter image (dmh_down_conv_*_img) against the dword kernels: bit-identical results, and the
time of both, at the encoder's three shapes, a window shape and the small-map cases (development tool).

    python3 tools/down_wide_check.py [batch=12]
"""
import sys

import torch

sys.path.insert(0, ".")
from depthmodelhardening_amd import _native as N  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda")
lib = N.lib()


def timeit(fn, it=20):
    fn()
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


def image(w3, wd, rows, inner):
    img = torch.empty(lib.dmh_down_conv_image_size(rows, inner), device=dev)
    N.check(lib.dmh_down_conv_weight_image(N.ptr(w3), N.ptr(wd), rows, inner, N.ptr(img), N.stream()))
    return img


ok = True
for (b, Ci, Co, H, W) in [(B, 64, 128, 80, 256), (B, 128, 256, 40, 128), (B, 256, 512, 20, 64), (B, 64, 128, 80, 112),
                          (2, 256, 512, 20, 64), (3, 128, 256, 36, 72), (32, 64, 128, 80, 256)]:
    g = torch.Generator(device="cuda").manual_seed(Ci + H)
    x = torch.randn(b, Ci, H, W, device=dev, generator=g)
    w3 = torch.randn(Co, Ci, 3, 3, device=dev, generator=g) * 0.05
    wd = torch.randn(Co, Ci, device=dev, generator=g) * 0.1
    s3, sd = torch.randn(Co, device=dev, generator=g), torch.randn(Co, device=dev, generator=g)
    img = image(w3, wd, Co, Ci)
    for down in (True, False):
        y = [torch.empty(b, Co, H // 2, W // 2, device=dev) for _ in range(4)]
        wdp, ydp = (N.ptr(wd), lambda t: N.ptr(t)) if down else (None, lambda t: None)
        old = lambda: N.check(lib.dmh_down_conv_fwd_act(N.ptr(x), N.ptr(w3), wdp, N.ptr(s3), N.ptr(sd) if down else None, 1, b, Ci, Co,
                                                        H, W, N.ptr(y[0]), ydp(y[1]), N.stream()))      # noqa: E731
        new = lambda: N.check(lib.dmh_down_conv_fwd_img(N.ptr(x), N.ptr(img), int(down), N.ptr(s3), N.ptr(sd) if down else None, 1, b,
                                                        Ci, Co, H, W, N.ptr(y[2]), ydp(y[3]), N.stream()))     # noqa: E731
        t_old, t_new = timeit(old), timeit(new)
        same = torch.equal(y[0], y[2]) and (not down or torch.equal(y[1], y[3]))
        ok &= same
        print("fwd  %3d->%3d @%dx%d B=%d down=%d: dword %.1f us, wide %.1f us, identical %s" % (Ci, Co, H, W, b, down, t_old, t_new, same))
    # backward-data: transposed filters
    w3t, wdt = w3.transpose(0, 1).contiguous(), wd.t().contiguous()
    imgt = image(w3t, wdt, Ci, Co)
    g3 = torch.randn(b, Co, H // 2, W // 2, device=dev, generator=g)
    gd = torch.randn(b, Co, H // 2, W // 2, device=dev, generator=g)
    gadd = torch.randn(b, Ci, H, W, device=dev, generator=g)
    for down in (True, False):
        o = [torch.empty(b, Ci, H, W, device=dev) for _ in range(2)]
        old = lambda: N.check(lib.dmh_down_conv_bwd_data_acc(N.ptr(g3), N.ptr(gd) if down else None, N.ptr(w3t),
                                                             N.ptr(wdt) if down else None, N.ptr(gadd), b, Ci, Co, H, W, N.ptr(o[0]),
                                                             N.stream()))     # noqa: E731
        new = lambda: N.check(lib.dmh_down_conv_bwd_data_img(N.ptr(g3), N.ptr(gd) if down else None, N.ptr(imgt), N.ptr(gadd), b, Ci,
                                                             Co, H, W, N.ptr(o[1]), N.stream()))      # noqa: E731
        t_old, t_new = timeit(old), timeit(new)
        same = torch.equal(o[0], o[1])
        ok &= same
        print("bwd  %3d->%3d @%dx%d B=%d down=%d: dword %.1f us, wide %.1f us, identical %s" % (Ci, Co, H, W, b, down, t_old, t_new, same))
print("ALL IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
