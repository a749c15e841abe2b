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


def test_fp64_reference_itself_matches_a_cpu_float64_convolution():
    x, w = _data(1, 64, 64, 80, 256, 1)
    y_gpu = F.conv2d(x.double(), w.double(), None, 1, 1)
    y_cpu = F.conv2d(x.double().cpu(), w.double().cpu(), None, 1, 1)
    assert float((y_gpu.cpu() - y_cpu).abs().max()) <= 1e-12 * float(y_cpu.abs().max())


# (name, B, C, K, H, W, pad): attack-pass shapes at 12 scenes
FWD_SHAPES = [("K10 layer1 64->64 @80x256", 12, 64, 64, 80, 256, 1),
              ("K10 layer3 256->256 @20x64", 12, 256, 256, 20, 64, 1),
              ("K10 upconv(2,1) 128->64 @82x258 pad0", 12, 128, 64, 82, 258, 0),
              ("K17 upconv(1,1) 96->32 @162x514 pad0", 12, 96, 32, 162, 514, 0),
              ("K11 upconv(0,1) 16->16 @322x1026 pad0", 4, 16, 16, 322, 1026, 0)]


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
              ("K18 upconv(1,1) 96->32 @162x514 pad0", 12, 96, 32, 162, 514, 0),
              ("K16 upconv(0,1) 16->16 @322x1026 pad0", 8, 16, 16, 322, 1026, 0)]


@pytest.mark.parametrize("shape", WRW_SHAPES, ids=[s[0].split(" @")[0].replace(" ", "_") for s in WRW_SHAPES])
def test_weight_gradient_vs_fp64(shape):
    """The reduction over 10k-10M pixels per filter tap: the fp64 weight gradient is formed image by image (bounded memory)."""
    from depthmodelhardening_amd import ops
    name, B, C, K, H, W, pad = shape
    x, w = _data(B, C, K, H, W, 11)
    wg = w.clone().requires_grad_(True)
    y = ops.conv3x3(x, wg, None, pad)
    g = (torch.randn(y.shape, generator=torch.Generator().manual_seed(12)) / (y.shape[2] * y.shape[3]) ** 0.5).cuda()
    (gw,) = torch.autograd.grad(y, wg, g)
    gw64 = torch.zeros(K, C, 3, 3, dtype=torch.float64, device="cuda")
    for b in range(B):
        gw64 += torch.nn.grad.conv2d_weight(x[b:b + 1].double(), (K, C, 3, 3), g[b:b + 1].double(), 1, pad)
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
