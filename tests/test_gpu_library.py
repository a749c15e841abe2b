"""torch.ops.dmh.* (depthmodelhardening_amd/library.py): the registered operators give the bits of ops.py's autograd
Functions, pass torch.library.opcheck (schema, fake tensors, autograd registration), trace under fake tensors, and back the
stand-alone layers surface (MD2/layers.py SSIM / get_smooth_loss) with kernels."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    from depthmodelhardening_amd import _native as N, library, ops  # noqa: F401
    return N, ops, library


def _paste_case(dev, n=3):
    from depthmodelhardening_amd.my_utils import ori_H, ori_W
    from depthmodelhardening_amd.physicalTrans import PhysicalTrans
    from oracle import synth
    obj, pmask = synth.make_object()
    pt = PhysicalTrans(obj, pmask, None, (1, 3, ori_H, ori_W), dist_range=list(np.arange(5, 10, 0.2)))
    coeffs = torch.from_numpy(pt.coeffs_for([5.0, 7.2, 9.8][:n], [0, 15, -30][:n])).to(dev)
    scenes = synth.kitti_like(n, 3, 375, 1242, torch.Generator().manual_seed(3)).to(dev)
    return scenes, obj.to(dev), pmask.to(dev), coeffs, pt.l_pad, pt.t_pad


def test_registered_attack_ops_equal_the_autograd_functions():
    N, ops, library = _lib()
    dev = torch.device("cuda")
    scenes, obj, pmask, coeffs, l_pad, t_pad = _paste_case(dev)
    res = []
    for use_lib in (False, True):
        patch = obj.clone().requires_grad_(True)
        if use_lib:
            adv, m = torch.ops.dmh.eot_paste(scenes, patch, pmask, coeffs, l_pad, t_pad, 320, 1024, None)
            cost = torch.ops.dmh.masked_sq_mean(adv[:, :1] * 0.5, m)
        else:
            adv, m = ops.eot_paste(scenes, patch, pmask, coeffs, l_pad, t_pad, (320, 1024))
            cost = ops.masked_sq_mean(adv[:, :1] * 0.5, m)
        (g,) = torch.autograd.grad(cost, patch)
        step = (torch.ops.dmh.pgd_linf_step if use_lib else ops.pgd_linf_step)(patch.detach(), obj, g, 0.02, 0.1)
        res.append((adv.detach(), m, cost.detach(), g, step))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert float(res[0][3].abs().max()) > 0
    # L0 operators
    g0 = torch.Generator().manual_seed(5)
    pos = torch.rand(obj.shape, generator=g0).to(dev).requires_grad_(True)
    neg = torch.rand(obj.shape, generator=g0).to(dev).requires_grad_(True)
    a1, c1 = ops.l0_compose(obj, pos, neg, 1 / 255.0)
    a2, c2 = torch.ops.dmh.l0_compose(obj, pos, neg, 1 / 255.0, False)
    assert torch.equal(a1, a2) and torch.equal(c1, c2)
    m1, m2 = ops.l0_mask_cost(pos, neg), torch.ops.dmh.l0_mask_cost(pos, neg)
    g1 = torch.autograd.grad(a1.sum() + m1, [pos, neg])
    g2 = torch.autograd.grad(a2.sum() + m2, [pos, neg])
    assert torch.equal(m1, m2) and all(torch.equal(x, y) for x, y in zip(g1, g2))


def test_registered_fused_loss_equals_ops_and_traces_under_fake_tensors():
    N, ops, library = _lib()
    from oracle import synth
    from tests.util import to_dev
    B, H, W = 2, 64, 192
    inputs, disps = synth.make_loss_case(B, H, W, 5)
    d_in = to_dev(inputs)
    res = []
    for use_lib in (False, True):
        dd = [d.cuda().requires_grad_(True) for d in disps]
        args = (d_in[("color", 0, 0)], [d_in[("color", "s", 0)]], [d_in["stereo_T"]], d_in[("K", 0)], d_in[("inv_K", 0)], dd,
                [d_in[("color", 0, s)] for s in range(4)])
        if use_lib:
            fin, sel, _ = torch.ops.dmh.photo_smooth_loss(*args, 0.1, 100.0, N.VARIANT_MD2, True, False, 1e-3, N.NOISE_NONE, 0, 0)
            sel0 = ops.SelectionMaps(sel, 4)[0]
        else:
            out = ops.photometric_smooth_loss(*args, noise=None)
            fin, sel0 = out.fin, out.sel[0]
        fin[N.FIN_LOSS].backward()
        res.append((fin.detach(), sel0, [d.grad for d in dd]))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert all(torch.equal(a, b) for a, b in zip(res[0][2], res[1][2]))
    # fake-tensor propagation: shapes and dtypes without touching the device
    from torch._subclasses.fake_tensor import FakeTensorMode
    mode = FakeTensorMode()
    fk = lambda t: mode.from_tensor(t)      # noqa: E731
    fargs = (fk(args[0]), [fk(t) for t in args[1]], [fk(t) for t in args[2]], fk(args[3]), fk(args[4]),
             [fk(d.detach()) for d in args[5]], [fk(t) for t in args[6]])
    with mode:
        f_fin, f_sel, f_st = torch.ops.dmh.photo_smooth_loss(*fargs, 0.1, 100.0, N.VARIANT_MD2, True, False, 1e-3,
                                                             N.NOISE_NONE, 0, 0)
        assert tuple(f_fin.shape) == (N.FIN_SIZE,) and tuple(f_sel.shape) == (B, H, W) and f_sel.dtype == torch.uint8
        assert tuple(f_st.shape) == (4, B, 2)


def test_opcheck_of_the_small_operators():
    """torch.library.opcheck: schema, fake implementation against the real one, autograd registration and the AOT dispatch path."""
    N, ops, library = _lib()
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(2)
    x = torch.rand(2, 3, 16, 24, generator=g).to(dev)
    y = torch.rand(2, 3, 16, 24, generator=g).to(dev)
    tests = ("test_schema", "test_faketensor", "test_autograd_registration")
    torch.library.opcheck(torch.ops.dmh.ssim_map, (x.clone().requires_grad_(True), y.clone().requires_grad_(True)), test_utils=tests)
    d = torch.rand(2, 1, 16, 24, generator=g).to(dev)
    torch.library.opcheck(torch.ops.dmh.smooth_loss, (d.clone().requires_grad_(True), x), test_utils=tests)
    torch.library.opcheck(torch.ops.dmh.masked_sq_mean, (d.clone().requires_grad_(True), (d > 0.5).float()), test_utils=tests)
    torch.library.opcheck(torch.ops.dmh.pgd_linf_step, (x, y, x - y, 0.02, 0.1), test_utils=tests)
    # --gt_depth term (K6b): same bits as ops.gt_depth_mse, mask passed as the dataset passes it (one channel expanded to three)
    m3 = (torch.rand(2, 1, 16, 24, generator=g) > 0.6).float().to(dev).expand(-1, 3, -1, -1)
    od = torch.tensor([[6.2], [8.4]], device=dev)
    d2 = (d * 0.3 + 0.01)
    torch.library.opcheck(torch.ops.dmh.gt_depth_mse, (d2.clone().requires_grad_(True), (d2 * 0.9).contiguous(), m3, od, 0.1, 100.0),
                          test_utils=tests)
    a, b = d2.clone().requires_grad_(True), d2.clone().requires_grad_(True)
    la, lb = torch.ops.dmh.gt_depth_mse(a, (d2 * 0.9).contiguous(), m3, od, 0.1, 100.0), ops.gt_depth_mse(b, d2 * 0.9, m3, od, 0.1, 100.0)
    la.backward()
    lb.backward()
    assert torch.equal(la, lb) and torch.equal(a.grad, b.grad)
    assert set(library.OPS) <= set(dir(torch.ops.dmh)) or all(hasattr(torch.ops.dmh, n) for n in library.OPS)


@pytest.mark.parametrize("shape", [(2, 3, 32, 96), (1, 3, 5, 7), (2, 1, 2, 2)])
def test_layers_ssim_goes_through_the_kernel_and_matches_the_formula(shape):
    """depthmodelhardening_amd.layers.SSIM on CUDA tensors = torch.ops.dmh.ssim_map: values and both input gradients against
    the reference's formula (MD2/layers.py:223-253) evaluated in float64 -- including 2-pixel-wide planes, where every
    window reads reflected pixels twice."""
    from depthmodelhardening_amd import layers
    g = torch.Generator().manual_seed(shape[2])
    x = torch.rand(shape, generator=g)
    y = (0.7 * x + 0.3 * torch.rand(shape, generator=g))
    xd, yd = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    out = layers.SSIM()(xd, yd)
    w = torch.rand(shape, generator=g)
    (out * w.cuda()).sum().backward()
    x64, y64 = x.double().requires_grad_(True), y.double().requires_grad_(True)
    xp, yp = F.pad(x64, (1, 1, 1, 1), mode="reflect"), F.pad(y64, (1, 1, 1, 1), mode="reflect")
    mu_x, mu_y = F.avg_pool2d(xp, 3, 1), F.avg_pool2d(yp, 3, 1)
    sx, sy = F.avg_pool2d(xp ** 2, 3, 1) - mu_x ** 2, F.avg_pool2d(yp ** 2, 3, 1) - mu_y ** 2
    sxy = F.avg_pool2d(xp * yp, 3, 1) - mu_x * mu_y
    ref = torch.clamp((1 - (2 * mu_x * mu_y + 1e-4) * (2 * sxy + 9e-4) / ((mu_x ** 2 + mu_y ** 2 + 1e-4) * (sx + sy + 9e-4))) / 2, 0, 1)
    (ref * w.double()).sum().backward()
    assert float((out.detach().cpu().double() - ref.detach()).abs().max()) <= 5e-5      # sigma = E[x^2] - mu^2 cancels to ~1e-3
    for got, want in ((xd.grad, x64.grad), (yd.grad, y64.grad)):
        err = float((got.cpu().double() - want).norm() / want.norm())
        assert err <= 2e-3, err                                                         # the same cancellation, differentiated


@pytest.mark.parametrize("shape", [(2, 3, 32, 96), (1, 3, 2, 5)])
def test_layers_smooth_loss_goes_through_the_kernel(shape):
    from depthmodelhardening_amd import layers
    B, Cc, H, W = shape
    g = torch.Generator().manual_seed(W)
    disp, img = torch.rand(B, 1, H, W, generator=g), torch.rand(B, Cc, H, W, generator=g)
    dd = disp.cuda().requires_grad_(True)
    out = layers.get_smooth_loss(dd, img.cuda())
    out.backward()
    d64, i64 = disp.double().requires_grad_(True), img.double()
    gdx, gdy = (d64[:, :, :, :-1] - d64[:, :, :, 1:]).abs(), (d64[:, :, :-1, :] - d64[:, :, 1:, :]).abs()
    gix = (i64[:, :, :, :-1] - i64[:, :, :, 1:]).abs().mean(1, keepdim=True)
    giy = (i64[:, :, :-1, :] - i64[:, :, 1:, :]).abs().mean(1, keepdim=True)
    ref = (gdx * torch.exp(-gix)).mean() + (gdy * torch.exp(-giy)).mean()
    ref.backward()
    assert abs(float(out) - float(ref)) <= 2e-6 * abs(float(ref))
    assert float((dd.grad.cpu().double() - d64.grad).abs().max()) <= 1e-5 * float(d64.grad.abs().max())
