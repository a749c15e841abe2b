"""fp64-anchored checks of the convolution kernels at the shapes they run at (K10, K11, K15, K16, K17, K18).

The reference runs these layers as torch.nn.Conv2d (MD2/layers.py:121-136 Conv3x3, torchvision's BasicBlock under
MD2/networks/resnet_encoder.py:85-98), i.e. as whatever fp32 library kernel the platform picks; an fp32-vs-fp32 comparison
with that library cannot say which side is closer to the mathematical convolution.  Here both are measured against the same
convolution in float64 (ATen's fp64 path: im2col + double GEMM, a different algorithm from every kernel under test, itself
cross-checked against a CPU float64 convolution on one image), at the real layer shapes of the attack pass (batch 12) and
the train pass (batch 32):

    rel-L2(HIP vs fp64)  <=  1.5 x rel-L2(library fp32 vs fp64) + 1e-7

Winograd F(2x2,3x3) kernels (K10 / K17 / K18) are in the same numerics class as MIOpen's own F(2,3) and must be as close to
exact arithmetic as the library is.  The direct MFMA kernels (K11 / K15 / K16) sum an output's products in ONE accumulator
chain where the library's implicit GEMMs split it: they get the chain's own rounding (0.5 sqrt(n) 2^-24) beside that bound.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, ref64):
    return float((a.double() - ref64).norm() / ref64.norm())


def _bound(name, e_hip, e_lib, chain=0):
    """``chain``: length of the kernel's single fp32 accumulation chain per output (the direct MFMA kernels K11 / K15 / K16
    sum all C x 9 products of an output into ONE accumulator; the library's implicit GEMMs split that sum): a chain of n
    rounded additions carries ~ 0.5 sqrt(n) 2^-24 of relative error (random-walk estimate), which is allowed beside the
    library-relative bound -- 1.4e-6 for layer4.0's 2,304 products, two orders below north_star's 1e-4."""
    print("%-44s rel-L2 vs fp64: hip %.3g  library fp32 %.3g" % (name, e_hip, e_lib))
    assert e_hip <= max(1.5 * e_lib + 1e-7, 0.5 * chain ** 0.5 * 2.0 ** -24), (name, e_hip, e_lib)


def _data(B, C, K, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g).cuda()
    w = (torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda()
    return x, w


def _wgrad64(x, g, pad):
    """Float64 weight gradient of a 3x3 stride-1 convolution, tap by tap: dw[k, c, ky, kx] = sum_{b, y, x} g[b, k, y, x] *
    x_pad[b, c, y + ky, x + kx] -- nine float64 contractions (GEMMs) instead of ATen's image-by-image float64
    convolution_backward, which took 10-14 s per full-resolution case; held to it below."""
    xp = F.pad(x.double(), (pad, pad, pad, pad))
    g64 = g.double()
    Ho, Wo = g.shape[2], g.shape[3]
    out = torch.empty(g.shape[1], x.shape[1], 3, 3, dtype=torch.float64, device=x.device)
    for ky in range(3):
        for kx in range(3):
            out[:, :, ky, kx] = torch.einsum("bkyx,bcyx->kc", g64, xp[:, :, ky:ky + Ho, kx:kx + Wo])
    return out


def test_fp64_reference_itself_matches_a_cpu_float64_convolution():
    x, w = _data(1, 64, 64, 80, 256, 1)
    y_gpu = F.conv2d(x.double(), w.double(), None, 1, 1)
    y_cpu = F.conv2d(x.double().cpu(), w.double().cpu(), None, 1, 1)
    assert float((y_gpu.cpu() - y_cpu).abs().max()) <= 1e-12 * float(y_cpu.abs().max())
    # the tap-by-tap float64 weight gradient against ATen's float64 convolution_backward on the CPU, both paddings
    for pad in (0, 1):
        xs, _ = _data(2, 24, 16, 18, 34, 3)
        gs = torch.randn(2, 16, 18 + 2 * pad - 2, 34 + 2 * pad - 2, generator=torch.Generator().manual_seed(4)).cuda()
        ref = torch.nn.grad.conv2d_weight(xs.double().cpu(), (16, 24, 3, 3), gs.double().cpu(), 1, pad)
        got = _wgrad64(xs, gs, pad).cpu()
        assert float((got - ref).abs().max()) <= 1e-12 * float(ref.abs().max())


# (name, B, C, K, H, W, pad): attack-pass shapes at 12 scenes
FWD_SHAPES = [("K10 layer1 64->64 @80x256", 12, 64, 64, 80, 256, 1),
              ("K10 layer3 256->256 @20x64", 12, 256, 256, 20, 64, 1),
              ("K10 upconv(2,1) 128->64 @82x258 pad0", 12, 128, 64, 82, 258, 0),
              ("K17 upconv(1,1) 96->32 @162x514 pad0", 12, 96, 32, 162, 514, 0),
              ("K11 upconv(0,1) 16->16 @322x1026 pad0", 4, 16, 16, 322, 1026, 0),
              # stream-K launches (round 5): 60 regions (MIOpen until then), 144 regions = 288 half items on 256 CUs, 54 regions
              ("K10 upconv(4,0) 512->256 @12x34 pad0 stream-K", 12, 512, 256, 12, 34, 0),
              ("K10 upconv(4,1) window 512->256 @22x30 pad0 stream-K", 12, 512, 256, 22, 30, 0),
              ("K10 upconv(3,0) window 256->128 @20x26 pad0 stream-K", 12, 256, 128, 20, 26, 0),
              # the strong-scaling share (2 attack scenes): 40 / 20 / 10 tile regions, library convolutions until the fill
              # threshold went to 64 work items or 512 stream-K units
              ("K10 layer3 256->256 @20x64 2 scenes", 2, 256, 256, 20, 64, 1),
              ("K10 layer4 512->512 @10x32 2 scenes", 2, 512, 512, 10, 32, 1),
              ("K10 upconv(4,0) 512->256 @12x34 pad0 2 scenes", 2, 512, 256, 12, 34, 0)]


@pytest.mark.parametrize("shape", FWD_SHAPES, ids=[s[0].split(" @")[0].replace(" ", "_") for s in FWD_SHAPES])
def test_forward_and_backward_data_vs_fp64(shape):
    from depthmodelhardening_amd import ops
    name, B, C, K, H, W, pad = shape
    x, w = _data(B, C, K, H, W, 7)
    y64 = F.conv2d(x.double(), w.double(), None, 1, pad)
    xg = x.clone().requires_grad_(True)
    with ops.frozen_weights():      # the attack pass: constant weights, backward-data only
        y = ops.conv3x3(xg, w, None, pad)
        g = torch.randn(y.shape, generator=torch.Generator().manual_seed(8)).cuda()
        (gx,) = torch.autograd.grad(y, xg, g)
    y_lib = torch.conv2d(x, w, None, 1, pad)
    _bound(name + " forward", _rel(y, y64), _rel(y_lib, y64))
    gx64 = F.conv_transpose2d(g.double(), w.double(), None, 1, pad)
    gx_lib = torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1,
                                                 [True, False, False])[0]
    _bound(name + " backward-data", _rel(gx, gx64), _rel(gx_lib, gx64))


# train-pass weight gradients: batch 32 where the maps are small; the two full-resolution layers at batch 12 / 8 (1M / 2.6M
# pixels per filter tap already; the float64 reference of the whole batch takes a minute)
WRW_SHAPES = [("K18 layer4 512->512 @10x32", 32, 512, 512, 10, 32, 1),
              ("K18 layer1 64->64 @80x256", 32, 64, 64, 80, 256, 1),
              ("K18 upconv(1,1) 96->32 @162x514 pad0", 8, 96, 32, 162, 514, 0),      # 8 / 4 images: the float64 reference is the cost
              ("K16 upconv(0,1) 16->16 @322x1026 pad0", 4, 16, 16, 322, 1026, 0)]


@pytest.mark.parametrize("shape", WRW_SHAPES, ids=[s[0].split(" @")[0].replace(" ", "_") for s in WRW_SHAPES])
def test_weight_gradient_vs_fp64(shape):
    """The reduction over 10k-10M pixels per filter tap, against the float64 sum taken tap by tap (_wgrad64)."""
    from depthmodelhardening_amd import ops
    name, B, C, K, H, W, pad = shape
    x, w = _data(B, C, K, H, W, 11)
    wg = w.clone().requires_grad_(True)
    y = ops.conv3x3(x, wg, None, pad)
    g = (torch.randn(y.shape, generator=torch.Generator().manual_seed(12)) / (y.shape[2] * y.shape[3]) ** 0.5).cuda()
    (gw,) = torch.autograd.grad(y, wg, g)
    gw64 = _wgrad64(x, g, pad)
    gw_lib = torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1,
                                                 [False, True, False])[1]
    _bound(name + " weight gradient", _rel(gw, gw64), _rel(gw_lib, gw64))


@pytest.mark.parametrize("shape", [(12, 64, 128, 80, 256), (12, 256, 512, 20, 64)], ids=["layer2.0", "layer4.0"])
def test_strided_block_entry_vs_fp64(shape):
    """K15: the 3x3 stride-2 convolution and the 1x1 stride-2 shortcut of a down-sampling block, forward and backward-data."""
    from depthmodelhardening_amd import ops
    B, C, K, H, W = shape
    g0 = torch.Generator().manual_seed(21)
    x = torch.randn(B, C, H, W, generator=g0).cuda()
    w3 = (torch.randn(K, C, 3, 3, generator=g0) * (2.0 / (9 * C)) ** 0.5).cuda()
    wd = (torch.randn(K, C, 1, 1, generator=g0) * (2.0 / C) ** 0.5).cuda()
    xg = x.clone().requires_grad_(True)
    with ops.frozen_weights():
        y3, yd = ops.down_convs(xg, w3, wd)
        g3 = torch.randn(y3.shape, generator=g0).cuda()
        gd = torch.randn(yd.shape, generator=g0).cuda()
        (gx,) = torch.autograd.grad([y3, yd], xg, [g3, gd])
    y3_64, yd_64 = F.conv2d(x.double(), w3.double(), None, 2, 1), F.conv2d(x.double(), wd.double(), None, 2, 0)
    _bound("K15 %d->%d 3x3/2 forward" % (C, K), _rel(y3, y3_64), _rel(torch.conv2d(x, w3, None, 2, 1), y3_64), chain=9 * C)
    _bound("K15 %d->%d 1x1/2 forward" % (C, K), _rel(yd, yd_64), _rel(torch.conv2d(x, wd, None, 2, 0), yd_64), chain=C)
    gx64 = (F.conv_transpose2d(g3.double(), w3.double(), None, 2, 1, output_padding=1) +
            F.conv_transpose2d(gd.double(), wd.double(), None, 2, 0, output_padding=1))
    lib = (torch.ops.aten.convolution_backward(g3, x, w3, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0] +
           torch.ops.aten.convolution_backward(gd, x, wd, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0])
    _bound("K15 %d->%d backward-data" % (C, K), _rel(gx, gx64), _rel(lib, gx64), chain=10 * K // 4)


def _wrw_chain(pixels, parts):
    """Accumulation structure of K20 / K21: ``parts`` partial sums (one per workgroup or wave) of pixels / parts products each,
    added in order: two chains, the random-walk estimate of each added in quadrature."""
    return pixels / parts + parts


# batch 32: the train pass's shapes
@pytest.mark.parametrize("shape", [(32, 64, 128, 80, 256), (32, 128, 256, 40, 128), (32, 256, 512, 20, 64), (3, 64, 64, 6, 24)],
                         ids=["layer2.0", "layer3.0", "layer4.0", "ragged"])
def test_strided_block_entry_weight_gradients_vs_fp64(shape):
    """K20: both weight gradients of a down-sampling block's entry (3x3 stride 2 and the 1x1 stride-2 shortcut) in one launch,
    against float64 and MIOpen, and bit for bit the same on a second run (no atomics)."""
    from depthmodelhardening_amd import ops
    B, C, K, H, W = shape
    g0 = torch.Generator().manual_seed(31)
    x = torch.randn(B, C, H, W, generator=g0).cuda()
    w3 = (torch.randn(K, C, 3, 3, generator=g0) * (2.0 / (9 * C)) ** 0.5).cuda().requires_grad_(True)
    wd = (torch.randn(K, C, 1, 1, generator=g0) * (2.0 / C) ** 0.5).cuda().requires_grad_(True)
    y3, yd = ops.down_convs(x, w3, wd)
    px = y3.shape[2] * y3.shape[3]
    g3 = (torch.randn(y3.shape, generator=g0) / px ** 0.5).cuda()
    gd = (torch.randn(yd.shape, generator=g0) / px ** 0.5).cuda()
    lib = __import__("depthmodelhardening_amd._native", fromlist=["x"]).lib()
    assert lib.dmh_down_wrw_workspace_size(B, C, K, H, W) > 0, "the shape must take K20, not the ATen fallback"
    gw3, gwd = torch.autograd.grad([y3, yd], [w3, wd], [g3, gd], retain_graph=True)
    gw3_2, gwd_2 = torch.autograd.grad([y3, yd], [w3, wd], [g3, gd])
    assert torch.equal(gw3, gw3_2) and torch.equal(gwd, gwd_2), "fixed-order sums: a second run must agree bit for bit"
    gw3_64 = torch.zeros(K, C, 3, 3, dtype=torch.float64, device="cuda")
    gwd_64 = torch.zeros(K, C, 1, 1, dtype=torch.float64, device="cuda")
    for b in range(B):
        gw3_64 += torch.nn.grad.conv2d_weight(x[b:b + 1].double(), (K, C, 3, 3), g3[b:b + 1].double(), 2, 1)
        gwd_64 += torch.nn.grad.conv2d_weight(x[b:b + 1].double(), (K, C, 1, 1), gd[b:b + 1].double(), 2, 0)
    l3 = torch.ops.aten.convolution_backward(g3, x, w3, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    ld = torch.ops.aten.convolution_backward(gd, x, wd, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    parts = min(512 // ((K // 64) * (C // 64)), B * (H // 2) * ((W // 2 + 31) // 32))
    chain = _wrw_chain(B * px, parts)
    _bound("K20 %d->%d 3x3/2 weight gradient" % (C, K), _rel(gw3, gw3_64), _rel(l3, gw3_64), chain=chain)
    _bound("K20 %d->%d 1x1/2 weight gradient" % (C, K), _rel(gwd, gwd_64), _rel(ld, gwd_64), chain=chain)
    # the 3x3 filter alone (a caller that holds the shortcut constant)
    y3, yd = ops.down_convs(x, w3, wd.detach())
    (only3,) = torch.autograd.grad([y3, yd], [w3], [g3, gd])
    assert torch.equal(only3, gw3)


@pytest.mark.parametrize("shape", [(8, 320, 1024), (2, 64, 192), (3, 38, 72)], ids=["kitti", "small", "ragged"])
def test_stem_weight_gradient_vs_fp64(shape):
    """K21: dW of conv1((x - 0.45) / 0.225) -- 3 -> 64 channels, 7x7, stride 2 -- against float64 and MIOpen on the normalised
    image, twice bit for bit; and the un-normalised form (stem_conv) through the same kernel."""
    from depthmodelhardening_amd import ops
    B, H, W = shape
    g0 = torch.Generator().manual_seed(41)
    x = torch.rand(B, 3, H, W, generator=g0).cuda()
    w = (torch.randn(64, 3, 7, 7, generator=g0) * (2.0 / 147) ** 0.5).cuda().requires_grad_(True)
    y = ops.stem_conv_norm(x, w, 0.45, 0.225)
    px = y.shape[2] * y.shape[3]
    g = (torch.randn(y.shape, generator=g0) / px ** 0.5).cuda()
    lib = __import__("depthmodelhardening_amd._native", fromlist=["x"]).lib()
    assert lib.dmh_stem_wrw_workspace_size(B, H, W) > 0
    (gw,) = torch.autograd.grad(y, w, g, retain_graph=True)
    (gw2,) = torch.autograd.grad(y, w, g)
    assert torch.equal(gw, gw2), "fixed-order sums: a second run must agree bit for bit"
    xn = (x - 0.45) / 0.225
    gw64 = torch.zeros(64, 3, 7, 7, dtype=torch.float64, device="cuda")
    for b in range(B):
        gw64 += torch.nn.grad.conv2d_weight(((x[b:b + 1].double() - 0.45) / 0.225), (64, 3, 7, 7), g[b:b + 1].double(), 2, 3)
    glib = torch.ops.aten.convolution_backward(g, xn, w, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    tiles = B * ((H // 2 + 3) // 4) * ((W // 2 + 31) // 32)
    _bound("K21 stem weight gradient %dx%dx%d" % shape, _rel(gw, gw64), _rel(glib, gw64),
           chain=_wrw_chain(B * px, 4 * min(tiles, 512)))
    # stem_conv: the caller normalised the image itself (mean 0, std 1 inside the kernel)
    xr = xn.clone().requires_grad_(True)
    y2 = ops.stem_conv(xr, w)
    (gw3,) = torch.autograd.grad(y2, w, g)
    assert _rel(gw3, gw64) <= 2 * _rel(gw, gw64) + 1e-7


def test_k15_image_form_is_bit_identical_to_the_dword_kernels():
    """K15 with 16-byte loaders and the filter image (dmh_down_conv_fwd_img / dmh_down_conv_bwd_data_img, round 5) runs the
    MFMAs of the dword kernels in the same order: equal bit for bit -- with and without the shortcut convolution, with the
    fused shift / ReLU and the epilogue addend, at the encoder's shapes, a window, a two-scene batch (32-channel tiles in the
    backward pass) and ragged tiles.  ops.py takes the image form wherever the rows are whole 16-byte words."""
    from depthmodelhardening_amd import _native as N
    lib = N.lib()
    dev = torch.device("cuda")

    def image(w3, wd, rows, inner):
        img = torch.empty(lib.dmh_down_conv_image_size(rows, inner), device=dev)
        N.check(lib.dmh_down_conv_weight_image(N.ptr(w3), N.ptr(wd), rows, inner, N.ptr(img), N.stream()))
        return img

    for (b, Ci, Co, H, W) in [(12, 64, 128, 80, 256), (12, 256, 512, 20, 64), (12, 64, 128, 80, 112), (2, 256, 512, 20, 64),
                              (3, 128, 256, 36, 72), (5, 64, 64, 10, 40)]:
        g = torch.Generator(device="cuda").manual_seed(Ci + H)
        x = torch.randn(b, Ci, H, W, device=dev, generator=g)
        w3 = torch.randn(Co, Ci, 3, 3, device=dev, generator=g) * 0.05
        wd = torch.randn(Co, Ci, device=dev, generator=g) * 0.1
        s3, sd = torch.randn(Co, device=dev, generator=g), torch.randn(Co, device=dev, generator=g)
        img = image(w3, wd, Co, Ci)
        for down in (True, False):
            y = [torch.zeros(b, Co, H // 2, W // 2, device=dev) for _ in range(4)]
            N.check(lib.dmh_down_conv_fwd_act(N.ptr(x), N.ptr(w3), N.ptr(wd) if down else None, N.ptr(s3), N.ptr(sd) if down else None,
                                              1, b, Ci, Co, H, W, N.ptr(y[0]), N.ptr(y[1]) if down else None, N.stream()))
            N.check(lib.dmh_down_conv_fwd_img(N.ptr(x), N.ptr(img), int(down), N.ptr(s3), N.ptr(sd) if down else None, 1, b, Ci, Co,
                                              H, W, N.ptr(y[2]), N.ptr(y[3]) if down else None, N.stream()))
            assert torch.equal(y[0], y[2]) and torch.equal(y[1], y[3]), ("forward", b, Ci, Co, H, W, down)
        ref = F.conv2d(x.double(), w3.double(), None, 2, 1) + s3.double().view(1, -1, 1, 1)
        assert _rel(y[2], ref.clamp_min(0)) < 2e-6
        w3t, wdt = w3.transpose(0, 1).contiguous(), wd.t().contiguous()
        imgt = image(w3t, wdt, Ci, Co)
        g3 = torch.randn(b, Co, H // 2, W // 2, device=dev, generator=g)
        gd = torch.randn(b, Co, H // 2, W // 2, device=dev, generator=g)
        gadd = torch.randn(b, Ci, H, W, device=dev, generator=g)
        for down in (True, False):
            o = [torch.zeros(b, Ci, H, W, device=dev) for _ in range(2)]
            N.check(lib.dmh_down_conv_bwd_data_acc(N.ptr(g3), N.ptr(gd) if down else None, N.ptr(w3t), N.ptr(wdt) if down else None,
                                                   N.ptr(gadd), b, Ci, Co, H, W, N.ptr(o[0]), N.stream()))
            N.check(lib.dmh_down_conv_bwd_data_img(N.ptr(g3), N.ptr(gd) if down else None, N.ptr(imgt), N.ptr(gadd), b, Ci, Co, H, W,
                                                   N.ptr(o[1]), N.stream()))
            assert torch.equal(o[0], o[1]), ("backward", b, Ci, Co, H, W, down)


def test_stream_k_is_deterministic_and_equals_whole_items():
    """Stream-K launches of the plain K10 path (csrc/wino_conv.hip): run twice bit for bit, with a bias, ragged tile regions
    included, and against the whole-item launch of the same kernel (DMH_WINO_SK switch) to fp32 re-association."""
    from depthmodelhardening_amd import ops
    for (B, C, K, H, W, pad) in ((12, 512, 256, 12, 34, 0), (12, 512, 256, 22, 30, 0), (12, 128, 256, 18, 24, 2),
                                 (12, 128, 64, 52, 66, 0), (5, 64, 128, 28, 36, 2), (12, 512, 512, 10, 32, 1)):
        x, w = _data(B, C, K, H, W, 21)
        b = torch.randn(K, generator=torch.Generator().manual_seed(22)).cuda()
        assert ops.WINO_SK
        with ops.frozen_weights():
            y1 = ops.conv3x3(x, w, b, pad)
            y2 = ops.conv3x3(x, w, b, pad)
            ops.WINO_SK = False
            try:
                y0 = ops.conv3x3(x, w, b, pad)      # whole items (2-way split with atomics, or MIOpen below the fill threshold)
            finally:
                ops.WINO_SK = True
        assert torch.equal(y1, y2), (B, C, K, H, W)
        y64 = F.conv2d(x.double(), w.double(), b.double(), 1, pad)
        e1, e0 = _rel(y1, y64), _rel(y0, y64)
        print("stream-K %s: rel-L2 vs fp64 %.3g (whole items / library %.3g)" % ((B, C, K, H, W, pad), e1, e0))
        assert e1 <= 1.5 * e0 + 1e-7, (e1, e0)


def test_stream_k_on_two_streams_side_by_side():
    """The partial work items of a stream-K launch live in a workspace the CALLER owns (ops._sk_workspace: one per device and
    stream).  Two streams that run decomposed convolutions at the same time give, bit for bit, what each gives alone."""
    from depthmodelhardening_amd import ops
    shapes = ((12, 512, 256, 12, 34, 0), (12, 512, 256, 22, 30, 0))
    data = [_data(*sh[:5], 31 + i) for i, sh in enumerate(shapes)]
    with ops.frozen_weights():
        alone = [ops.conv3x3(x, w, None, sh[5]) for (x, w), sh in zip(data, shapes)]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        outs = [[], []]
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
        for _ in range(6):
            for i, s in enumerate(streams):
                with torch.cuda.stream(s):
                    outs[i].append(ops.conv3x3(data[i][0], data[i][1], None, shapes[i][5]))
        torch.cuda.synchronize()
    assert len({k[1] for k in ops._sk_ws}) >= 3         # the default stream's workspace and one per side stream
    for i in range(2):
        assert all(torch.equal(o, alone[i]) for o in outs[i]), i


def test_stream_k_of_the_32_channel_kernel():
    """K17's stream-K form (csrc/wino32_conv.hip) through the C ABI at the attack's window shapes of upconv(1,1) / upconv(1,0): the
    same result as the whole-item launch up to fp32 re-association, against float64, run twice bit for bit."""
    from depthmodelhardening_amd import _native as N, ops
    lib = N.lib()
    ws = torch.empty(8 << 20, device="cuda")
    for (B, C, K, H, W, pad) in ((12, 96, 32, 94, 116, 0), (12, 64, 32, 50, 62, 0), (12, 32, 96, 92, 114, 2), (3, 96, 32, 30, 200, 0)):
        x, w = _data(B, C, K, H, W, 31)
        b = torch.randn(K, generator=torch.Generator().manual_seed(32)).cuda()
        U = torch.empty(lib.dmh_wino32_weight_size(K, C), device="cuda")
        N.check(lib.dmh_wino32_weight_transform(N.ptr(w), K, C, 0, N.ptr(U), N.stream()))
        Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
        ys = [torch.empty(B, K, Ho, Wo, device="cuda") for _ in range(3)]
        for y in ys[:2]:
            N.check(lib.dmh_wino32_conv3x3_ws(N.ptr(x), N.ptr(U), N.ptr(b), B, C, K, H, W, pad, N.ptr(y), N.ptr(ws), ws.numel(), N.stream()))
        N.check(lib.dmh_wino32_conv3x3(N.ptr(x), N.ptr(U), N.ptr(b), B, C, K, H, W, pad, N.ptr(ys[2]), N.stream()))
        assert torch.equal(ys[0], ys[1]), (B, C, K, H, W)
        y64 = F.conv2d(x.double(), w.double(), b.double(), 1, pad)
        e1, e0 = _rel(ys[0], y64), _rel(ys[2], y64)
        print("K17 stream-K %s: rel-L2 vs fp64 %.3g (whole items %.3g); differs from whole items: %s" % (
            (B, C, K, H, W, pad), e1, e0, not torch.equal(ys[0], ys[2])))
        assert e1 <= 1.5 * e0 + 1e-7, (e1, e0)

