"""The attack's cropped decoder tail (roi.py, csrc/roi_glue.hip, ops.roi_tail_cost).

CPU: the window plan covers what every stage reads, for every pose of the training grid.
GPU: the windowed glue pass against torch ops on the full frame; the windowed cost and its patch gradient against the
full-frame attack step (torchattacks/attacks/phy_obj_atk.py:87-97) on the real U-Net, over a spread of the 25 x 13 poses.
"""
import numpy as np
import pytest
import torch

from depthmodelhardening_amd.roi import LEVEL, STAGES, TABLE, WINDOWS, RoiPlan, mask_box

H, W = 320, 1024


def _pose_grid():
    from depthmodelhardening_amd.my_utils import ori_H, ori_W
    from depthmodelhardening_amd.physicalTrans import PhysicalTrans
    pt = PhysicalTrans(torch.zeros(1, 3, 260, 300), torch.ones(1, 1, 260, 300), None, (1, 3, ori_H, ori_W),
                       dist_range=list(np.arange(5, 10, 0.2)))
    return pt, [(z, a) for z in pt.dist_range for a in pt.angle_range]


def _covers(org_in, size_in, org_out, size_out, frame, halve):
    """window `in` (one axis) holds everything the stage producing window `out` reads"""
    lo = np.maximum(org_out - 1, 0)
    hi = np.minimum(org_out + size_out + 1, frame)
    if halve:
        lo, hi = lo >> 1, ((hi - 1) >> 1) + 1
    return bool(np.all(org_in <= lo) and np.all(org_in + size_in >= hi))


def _check_plan(plan, boxes, trial):
    """every window of ``plan`` holds what its consumer reads (the invariants the kernels rely on)"""
    for n in WINDOWS:
        fh, fw = H >> LEVEL[n], W >> LEVEL[n]
        hc, wc = plan.size[n]
        o = plan.org[n]
        assert hc % 2 == 0 and wc % 2 == 0 and 2 <= hc <= fh and 2 <= wc <= fw
        assert (o % 2 == 0).all() and (o >= 0).all() and (o[:, 0] + hc <= fh).all() and (o[:, 1] + wc <= fw).all()
    # the head's window holds the mask's box
    d = plan.org["d"]
    assert (d[:, 0] <= boxes[:, 0]).all() and (d[:, 0] + plan.size["d"][0] >= boxes[:, 1]).all()
    assert (d[:, 1] <= boxes[:, 2]).all() and (d[:, 1] + plan.size["d"][1] >= boxes[:, 3]).all()
    chain = [(STAGES[k + 1][0], STAGES[k][0], STAGES[k][2]) for k in range(len(STAGES) - 1)]
    assert chain[:3] == [("z01", "d", False), ("y00", "z01", True), ("z11", "y00", False)] and len(chain) == 9
    for src, dst, halve in chain:
        for ax, frame in ((0, H >> LEVEL[dst]), (1, W >> LEVEL[dst])):
            assert _covers(plan.org[src][:, ax], plan.size[src][ax], plan.org[dst][:, ax], plan.size[dst][ax], frame,
                           halve), (trial, src, dst, ax)
    assert plan.table().shape == (len(TABLE), len(boxes), 2) and plan.table().dtype == np.int32
    # regions of the whole-frame sources hold what the tail reads of them; the encoder head's windows hold what it reads
    for reg, dst, halve in (("r_y20", "z21", True), ("r_f1", "z21", False), ("r_f0", "z11", False), ("r_y30", "z31", True),
                            ("r_f2", "z31", False), ("r_y40", "z41", True), ("r_f3", "z41", False)):
        for ax, frame in ((0, H >> LEVEL[dst]), (1, W >> LEVEL[dst])):
            assert _covers(plan.org[reg][:, ax], plan.size[reg][ax], plan.org[dst][:, ax], plan.size[dst][ax], frame, halve)
    for ax, (fd, fs, fq) in enumerate(((H, H >> 1, H >> 2), (W, W >> 1, W >> 2))):
        d0, dn = plan.org["d"][:, ax].astype(int), plan.size["d"][ax]
        s0, sn = plan.org["gz"][:, ax].astype(int), plan.size["gz"][ax]
        q0, qn = plan.org["l1"][:, ax].astype(int), plan.size["l1"][ax]
        f0_, fn = plan.org["r_f0"][:, ax].astype(int), plan.size["r_f0"][ax]
        # conv1's adjoint reads rows Y-1 .. Y+2 of its output gradient for image rows 2Y, 2Y+1
        assert (s0 <= np.maximum((d0 >> 1) - 1, 0)).all() and (s0 + sn >= np.minimum(((d0 + dn) >> 1) + 2, fs)).all()
        # the max-pool adjoint reads cells r >> 1 .. (r + 1) >> 1; layer1 spoils four rings of its window
        assert (q0 <= np.maximum((s0 >> 1) - 4, 0)).all() and (q0 + qn >= np.minimum(((s0 + sn) >> 1) + 1 + 4, fq)).all()
        # the encoder head reads feature 0's gradient on "gz": inside the rectangle the tail writes
        assert (f0_ <= s0).all() and (f0_ + fn >= s0 + sn).all()
        # incremental head: the cells written into feature 1 ("f1s") lie inside the compact window "hl" minus the rings its
        # same-size stages spoil (pooling 1 + four convolutions: 5 at the top / left, 4 at the bottom / right; none at a
        # frame edge); the cells layer1's backward must deliver (under "gz") lie 9 rings inside it; "hz" = 2 x "hl" holds
        # "gz" and what the tail reads of feature 0
        h0, hn = plan.org["hl"][:, ax].astype(int), plan.size["hl"][ax]
        w0, wn = plan.org["f1s"][:, ax].astype(int), plan.size["f1s"][ax]
        assert (((w0 - h0) >= 5) | (h0 == 0)).all() and ((((h0 + hn) - (w0 + wn)) >= 4) | (h0 + hn == fq)).all()
        c0, c1 = s0 >> 1, ((s0 + sn) >> 1) + 1
        assert (((c0 - h0) >= 9) | (h0 == 0)).all() and ((((h0 + hn) - np.minimum(c1, fq)) >= 9) | (h0 + hn == fq)).all()
        z0_, zn = plan.org["hz"][:, ax].astype(int), plan.size["hz"][ax]
        assert (z0_ == 2 * h0).all() and zn == 2 * hn
        assert (z0_ <= s0).all() and (z0_ + zn >= s0 + sn).all() and (z0_ <= f0_).all() and (z0_ + zn >= f0_ + fn).all()
    assert plan.head_incremental_ok and plan.layer2_incremental_ok
    # layer2 on "h3": the cells written into feature 2 ("f2s") lie inside it minus the rings its stages spoil (stride-2
    # entry + three convolutions: 4 top / left, 3 bottom / right); the cells whose gradient the stride-2 adjoint reads to
    # cover "hl" lie 7 rings inside; "h3in" = 2 x "h3" holds "hl" at the even origin "hl_rel"
    for ax, (f8, f4) in enumerate(((H >> 3, H >> 2), (W >> 3, W >> 2))):
        c0, cn = plan.org["h3"][:, ax].astype(int), plan.size["h3"][ax]
        w0, wn = plan.org["f2s"][:, ax].astype(int), plan.size["f2s"][ax]
        assert (((w0 - c0) >= 4) | (c0 == 0)).all() and ((((c0 + cn) - (w0 + wn)) >= 3) | (c0 + cn == f8)).all()
        l0, ln = plan.org["hl"][:, ax].astype(int), plan.size["hl"][ax]
        q0, q1 = l0 >> 1, np.minimum(((l0 + ln) >> 1) + 1, f8)
        assert (((q0 - c0) >= 7) | (c0 == 0)).all() and ((((c0 + cn) - q1) >= 7) | (c0 + cn == f8)).all()
        r0 = plan.org["hl_rel"][:, ax].astype(int)
        assert (r0 == l0 - 2 * c0).all() and (r0 >= 0).all() and (r0 + ln <= 2 * cn).all() and (r0 % 2 == 0).all()
        assert (plan.org["h3in"][:, ax] == 2 * c0).all() and plan.size["h3in"][ax] == 2 * cn


def test_plan_covers_every_read_for_all_training_poses():
    pt, grid = _pose_grid()
    rng = np.random.RandomState(0)
    for trial in range(40):
        idx = rng.choice(len(grid), 12, replace=False)
        boxes = pt.mask_boxes([grid[i][0] for i in idx], [grid[i][1] for i in idx], (H, W))
        _check_plan(RoiPlan(boxes, H, W), boxes, trial)
    # the windows are a small part of the frame even for the nearest object
    near = RoiPlan(pt.mask_boxes([5.0], [0], (H, W)), H, W)
    assert near.area_fraction()["z01"] < 0.25 and near.area_fraction()["l1"] < 0.35


def test_common_size_plans_give_every_step_the_same_shapes_and_still_cover():
    """roi.common_size_plans: the steps of an attack (new poses at every step) get ONE set of window sizes -- the condition for
    replaying a captured HIP graph -- and every plan keeps every coverage invariant; the sizes are the element-wise maximum of
    the steps' own sizes or slightly above (a larger window makes the windows derived from it larger), never below."""
    from depthmodelhardening_amd.roi import common_size_plans
    pt, grid = _pose_grid()
    rng = np.random.RandomState(1)
    for trial, (nsteps, nscenes, depth) in enumerate(((10, 12, 4), (10, 2, 4), (20, 12, 3), (3, 1, 2), (10, 12, 4))):
        steps = []
        for _ in range(nsteps):
            idx = rng.choice(len(grid), nscenes, replace=False)
            steps.append(pt.mask_boxes([grid[i][0] for i in idx], [grid[i][1] for i in idx], (H, W)))
        plans = common_size_plans(steps, H, W, depth=depth)
        assert plans is not None and len(plans) == nsteps
        own = [RoiPlan(b, H, W, depth=depth) for b in steps]
        for n in TABLE:
            assert len({p.size[n] for p in plans}) == 1
            top = (max(p.size[n][0] for p in own), max(p.size[n][1] for p in own))
            assert plans[0].size[n][0] >= top[0] and plans[0].size[n][1] >= top[1]
            assert plans[0].size[n][0] <= top[0] + 16 and plans[0].size[n][1] <= top[1] + 16, (n, plans[0].size[n], top)
        for p, b in zip(plans, steps):
            _check_plan(p, b, trial)
    # one step: its own plan
    one = common_size_plans(steps[:1], H, W, depth=2)
    assert one[0].size == RoiPlan(steps[0], H, W, depth=2).size


def test_mask_box_clips_and_is_conservative():
    # a quad hanging over the bottom edge (the nearest poses do) is clipped to the frame
    y0, y1, x0, x1 = mask_box([[500, 200], [700, 200], [700, 420], [500, 420]], (375, 1242), (H, W))
    assert y1 == H and 0 <= y0 < y1 and 0 <= x0 < x1 <= W
    # output pixel o reads source rows (o + 0.5) * s - 0.5 and the next: every output pixel whose taps touch the quad lies inside
    s = 375.0 / H
    rows = [o for o in range(H) if np.floor((o + 0.5) * s - 0.5) + 1 >= 199 and np.floor((o + 0.5) * s - 0.5) <= 421]
    assert y0 <= rows[0] and rows[-1] < y1


def test_single_box_plan_and_small_frames():
    plan = RoiPlan([[10, 20, 30, 50]], 64, 96)
    assert plan.size["d"][0] >= 10 and plan.size["d"][1] >= 20
    whole = RoiPlan([[0, 64, 0, 96]], 64, 96)
    assert whole.size["d"] == (64, 96) and whole.size["y10"] == (16, 24) and not whole.table().any()
    with pytest.raises(RuntimeError):
        RoiPlan([[0, 4, 0, 4]], 20, 30)


# ---------------------------------------------------------------------------------------------------------------- GPU
def _pad_ref(y, skip, up, elu):
    import torch.nn.functional as F
    t = F.elu(y) if elu else y
    if up:
        t = F.interpolate(t, scale_factor=2, mode="nearest")
    if skip is not None:
        t = torch.cat([t, skip], 1)
    return F.pad(t, (1, 1, 1, 1), mode="reflect")


@pytest.mark.gpu
@pytest.mark.parametrize("up,with_skip,windowed_src", [(0, False, False), (0, False, True), (1, False, True), (1, True, True),
                                                        (1, True, False)])
def test_roi_glue_matches_the_full_frame_pass(up, with_skip, windowed_src):
    """out == the window of pad1_reflect(cat(up2(ELU(y)), skip)); backward == autograd through the same torch ops with the
    gradient embedded in the full padded frame -- including windows that touch all four borders."""
    import ctypes as C

    from depthmodelhardening_amd import _native as N, ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(3 + up)
    B, C1, C2 = 5, 6, (4 if with_skip else 0)
    FH, FW = 24, 40                                       # destination frame
    sh, sw = (FH // 2, FW // 2) if up else (FH, FW)       # y's frame
    hc, wc = 10, 12
    dst_org = torch.tensor([[0, 0], [FH - hc, FW - wc], [0, FW - wc], [6, 14], [FH - hc, 0]], dtype=torch.int32)
    y_full = torch.randn(B, C1, sh, sw, generator=g)
    skip_full = torch.randn(B, C2, FH, FW, generator=g) if with_skip else None
    ref_in_y = y_full.clone().requires_grad_(True)
    ref_in_s = skip_full.clone().requires_grad_(True) if with_skip else None
    ref_pad = _pad_ref(ref_in_y, ref_in_s, up, True)                      # [B, C, FH+2, FW+2]
    if windowed_src:        # y given as a window of its frame that holds everything the destination window reads
        ywh, yww = (8, 10) if up else (12, 14)
        lo = np.maximum(dst_org.numpy() - 1, 0)
        if up:
            lo = lo >> 1
        y_org = torch.from_numpy(np.minimum(lo, [sh - ywh, sw - yww]).astype(np.int32))
        y_in = torch.stack([y_full[b, :, y_org[b, 0]:y_org[b, 0] + ywh, y_org[b, 1]:y_org[b, 1] + yww] for b in range(B)])
    else:
        y_org, y_in = None, y_full
    yd = y_in.to(dev).contiguous()
    sd = skip_full.to(dev).contiguous() if with_skip else None
    org_d, y_org_d = dst_org.to(dev), (None if y_org is None else y_org.to(dev))   # the struct holds raw pointers: keep them alive
    a = ops._roi_glue_args(yd, y_org_d, sd, None, org_d, (hc, wc), (FH, FW), up, 1)
    out = ops._roi_glue_fwd(a, dev)
    want = torch.stack([ref_pad[b, :, dst_org[b, 0]:dst_org[b, 0] + hc + 2, dst_org[b, 1]:dst_org[b, 1] + wc + 2]
                        for b in range(B)])
    assert torch.allclose(out.cpu(), want, rtol=1e-6, atol=1e-6)
    # backward: a random gradient on the window, zero elsewhere in the padded frame
    g_out = torch.randn(out.shape, generator=g)
    g_full = torch.zeros_like(ref_pad)
    for b in range(B):
        g_full[b, :, dst_org[b, 0]:dst_org[b, 0] + hc + 2, dst_org[b, 1]:dst_org[b, 1] + wc + 2] = g_out[b]
    ref_pad.backward(g_full)
    g_y, g_skip = ops._roi_glue_bwd(a, g_out.to(dev).contiguous(), dev, with_skip)
    if windowed_src:
        want_y = torch.stack([ref_in_y.grad[b, :, y_org[b, 0]:y_org[b, 0] + ywh, y_org[b, 1]:y_org[b, 1] + yww]
                              for b in range(B)])
        # nothing of the gradient falls outside the source windows
        inside = torch.zeros_like(ref_in_y.grad)
        for b in range(B):
            inside[b, :, y_org[b, 0]:y_org[b, 0] + ywh, y_org[b, 1]:y_org[b, 1] + yww] = want_y[b]
        assert torch.equal(inside, ref_in_y.grad)
    else:
        want_y = ref_in_y.grad
    assert torch.allclose(g_y.cpu(), want_y, rtol=1e-5, atol=1e-6)
    if with_skip:
        assert torch.allclose(g_skip.cpu(), ref_in_s.grad, rtol=1e-6, atol=1e-6)
    assert C.sizeof(N.RoiGlueArgs) == 5 * 8 + 13 * 4 + 4       # five pointers, thirteen ints, tail padding


def _unet(dev, seed=0):
    from depthmodelhardening_amd.depth_model import import_depth_model
    torch.manual_seed(seed)
    model = import_depth_model((1024, 320)).to(dev).eval()
    # random-init BatchNorm statistics are (0, 1): perturb them so that the eval-mode affine is not the identity
    g = torch.Generator().manual_seed(seed + 1)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(0.1 * torch.randn(m.num_features, generator=g))
            m.running_var.copy_(1 + 0.2 * torch.rand(m.num_features, generator=g))
    return model


@pytest.mark.gpu
def test_mask_is_zero_outside_the_boxes_for_all_training_poses():
    from depthmodelhardening_amd import ops
    from depthmodelhardening_amd.my_utils import to_device_async
    dev = torch.device("cuda")
    pt, grid = _pose_grid()
    patch = torch.rand(1, 3, 260, 300, device=dev)
    yy, xx = torch.meshgrid(torch.arange(260.), torch.arange(300.), indexing="ij")
    for pmask in (torch.ones(1, 1, 260, 300), (((yy - 130) / 110) ** 2 + ((xx - 150) / 140) ** 2 <= 1).float()[None, None]):
        pmask = pmask.to(dev)
        scene = torch.rand(1, 3, 375, 1242, device=dev)
        for lo in range(0, len(grid), 25):
            part = grid[lo:lo + 25]
            z0, al = [p[0] for p in part], [p[1] for p in part]
            coeffs = to_device_async(pt.coeffs_for(z0, al), dev)
            _, m = ops.eot_paste(scene, patch, pmask, coeffs, pt.l_pad, pt.t_pad, (H, W))
            boxes = pt.mask_boxes(z0, al, (H, W))
            inside = torch.zeros_like(m)
            for b, (y0, y1, x0, x1) in enumerate(boxes):
                inside[b, :, y0:y1, x0:x1] = 1
            assert float((m * (1 - inside)).abs().max()) == 0.0
            assert float(m.sum()) > 0


@pytest.mark.gpu
def test_windowed_attack_step_equals_the_full_frame_step():
    """cost and d cost / d patch of one attack step (phy_obj_atk.py:87-97): windows around the object vs the whole frame."""
    from depthmodelhardening_amd import ops
    from depthmodelhardening_amd.my_utils import to_device_async
    from oracle import synth
    dev = torch.device("cuda")
    model = _unet(dev)
    pt, grid = _pose_grid()
    obj, pmask = synth.make_object()
    obj, pmask = obj.to(dev), pmask.to(dev)
    scenes = synth.kitti_like(12, 3, 375, 1242, torch.Generator().manual_seed(5)).to(dev)
    rng = np.random.RandomState(4)
    worst_g = worst_c = 0.0
    with ops.frozen_weights():
        for trial in range(3):
            if trial == 0:      # the extremes: nearest / farthest, both yaw limits
                poses = [(5.0, 0), (5.0, -30), (5.0, 30), (9.8, 0), (9.8, 30), (9.8, -30), (7.0, 15), (6.2, -10),
                         (8.4, 5), (5.2, 20), (9.0, -25), (7.6, 0)]
            else:
                poses = [grid[i] for i in rng.choice(len(grid), 12, replace=False)]
            z0, al = [p[0] for p in poses], [p[1] for p in poses]
            coeffs = to_device_async(pt.coeffs_for(z0, al), dev)
            res = []
            for depth in (None, 2, 3, 4):
                plan = tab = None
                if depth is not None:
                    plan = RoiPlan(pt.mask_boxes(z0, al, (H, W)), H, W, depth=depth)
                    tab = to_device_async(plan.table(), dev)
                patch = obj.clone().requires_grad_(True)
                adv, m = ops.eot_paste(scenes, patch, pmask, coeffs, pt.l_pad, pt.t_pad, (H, W))
                cost = -(model.masked_sq_mean(adv, m, plan, tab) if plan is not None else ops.masked_sq_mean(model(adv), m))
                (grad,) = torch.autograd.grad(cost, patch)
                res.append((cost.detach().double().cpu(), grad.detach().double().cpu()))
                assert plan is None or plan.head_windowed
            c0, g0 = res[0]
            assert float(g0.abs().max()) > 0
            for c1, g1 in res[1:]:
                worst_c = max(worst_c, abs(float(c1 - c0)) / abs(float(c0)))
                worst_g = max(worst_g, float((g1 - g0).norm() / g0.norm()))
                # element-wise: the same convolution arithmetic on the same tiles, only the reduction of the cost differs
                assert float((g1 - g0).abs().max()) <= 2e-5 * float(g0.abs().max()), trial
    print("windowed vs full frame: cost rel %.3g, patch gradient rel-L2 %.3g" % (worst_c, worst_g))
    assert worst_c <= 1e-6 and worst_g <= 1e-6


@pytest.mark.gpu
def test_attack_with_windows_equals_attack_without():
    """A whole 3-step Phy_obj_atk on the U-Net: use_roi on / off give the same patch (sign steps of equal gradients)."""
    import random

    from depthmodelhardening_amd.torchattacks import Phy_obj_atk
    from oracle import synth
    dev = torch.device("cuda")
    model = _unet(dev, seed=2)
    obj, pmask = synth.make_object()
    scenes = synth.kitti_like(12, 3, 375, 1242, torch.Generator().manual_seed(8)).to(dev)
    noise = (torch.rand(obj.shape, generator=torch.Generator().manual_seed(9)) * 2 - 1) * 0.1
    out = []
    for use_roi in (False, True):
        atk = Phy_obj_atk(model, obj.to(dev), pmask.to(dev), eps=0.1, alpha=0.02, steps=3, dist_range=list(np.arange(5, 10, 0.2)))
        atk.use_roi = use_roi
        atk.random_start_noise = noise
        random.seed(13)
        adv, ben, m, patch = atk(scenes, 12)
        out.append((adv.cpu(), m.cpu(), patch.cpu()))
    assert torch.equal(out[0][1], out[1][1])
    agree = (out[0][2] == out[1][2]).float().mean().item()
    print("patch texels identical with / without windows: %.5f" % agree)
    assert agree > 0.999        # a sign() step on a ~0 gradient may flip a texel by 2 alpha


@pytest.mark.gpu
def test_attack_replayed_from_a_hip_graph_equals_the_eager_attack():
    """Phy_obj_atk.use_graph: common window sizes for all steps, step 0 eager, step 1 captured, steps 2 .. n-1 replayed.
    Bit for bit the eager attack on the same (common-size) windows -- the replays launch the very kernels of the eager loop --
    and, to the sign() flips of ~0 gradients, the default attack on each step's own windows.  A second attack (new graph, the
    first one's memory pool) with a different random start shows that nothing of the first is baked in."""
    import random

    from depthmodelhardening_amd.torchattacks import Phy_obj_atk
    from oracle import synth
    dev = torch.device("cuda")
    model = _unet(dev, seed=5)
    obj, pmask = synth.make_object()
    scenes = synth.kitti_like(12, 3, 375, 1242, torch.Generator().manual_seed(28)).to(dev)
    noises = [(torch.rand(obj.shape, generator=torch.Generator().manual_seed(29 + k)) * 2 - 1) * 0.1 for k in range(2)]
    out = {}
    for mode in ("own", "common", "graph"):
        atk = Phy_obj_atk(model, obj.to(dev), pmask.to(dev), eps=0.1, alpha=0.02, steps=5, dist_range=list(np.arange(5, 10, 0.2)))
        atk.common_windows, atk.use_graph = mode == "common", mode == "graph"
        for k, noise in enumerate(noises):
            atk.random_start_noise = noise
            random.seed(33 + k)
            adv, ben, m, patch = atk(scenes, 12)
            out[mode, k] = (adv.cpu(), patch.cpu())
        assert (atk._graph is not None) == (mode == "graph")
    for k in range(2):
        assert torch.equal(out["graph", k][1], out["common", k][1]) and torch.equal(out["graph", k][0], out["common", k][0])
        agree = (out["graph", k][1] == out["own", k][1]).float().mean().item()
        print("attack %d: patch texels identical, graph on common windows / eager on each step's own: %.5f" % (k, agree))
        assert agree > 0.999
    assert not torch.equal(out["graph", 0][1], out["graph", 1][1])


@pytest.mark.gpu
def test_a_failed_graph_capture_falls_back_to_eager_launches():
    """A HIP-graph capture that dies half way (an injected error after the captured step's first launch: what a HIP call of
    another thread does to a capture in the default error mode) must not cost the attack: nothing of the captured step has
    executed, so the eager loop takes over on the same common-size windows from step 1 -- the result is bit for bit the
    common-windows attack --, use_graph switches itself off with the reason kept, and the next attack runs eagerly."""
    import random
    import warnings

    from depthmodelhardening_amd.torchattacks import Phy_obj_atk
    from oracle import synth
    dev = torch.device("cuda")
    model = _unet(dev, seed=5)
    obj, pmask = synth.make_object()
    scenes = synth.kitti_like(4, 3, 375, 1242, torch.Generator().manual_seed(28)).to(dev)
    noise = (torch.rand(obj.shape, generator=torch.Generator().manual_seed(29)) * 2 - 1) * 0.1
    out = {}
    for mode in ("common", "broken"):
        atk = Phy_obj_atk(model, obj.to(dev), pmask.to(dev), eps=0.1, alpha=0.02, steps=4, dist_range=list(np.arange(5, 10, 0.2)))
        atk.common_windows, atk.use_graph, atk._capture_fault = mode == "common", mode == "broken", mode == "broken"
        atk.random_start_noise = noise
        random.seed(33)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            adv, ben, m, patch = atk(scenes, 4)
        out[mode] = (adv.cpu(), patch.cpu())
        if mode == "broken":
            assert atk.use_graph is False and atk._graph is None and "injected capture fault" in atk.graph_failure
            assert any("capture of the attack step failed" in str(w.message) for w in caught)
            atk.common_windows = True           # the next attack of the same object: eager, no capture attempted
            random.seed(33)
            adv2, _, _, patch2 = atk(scenes, 4)
            assert torch.equal(patch2.cpu(), out["common"][1])
    assert torch.equal(out["broken"][1], out["common"][1]) and torch.equal(out["broken"][0], out["common"][0])


@pytest.mark.gpu
def test_l0_attack_with_windows_equals_attack_without():
    """BASELINE config 3's timed path: Phy_obj_atk_l0 takes the same windowed cost (phy_obj_atk_l0.py:118-127 through
    DepthModelWrapper.masked_sq_mean).  use_roi on / off: the same per-iteration trace, the same pattern gradients handed to
    Adam in every iteration, the same patterns and patch afterwards."""
    import random

    from depthmodelhardening_amd.torchattacks import Phy_obj_atk_l0
    from oracle import synth
    dev = torch.device("cuda")
    model = _unet(dev, seed=4)
    obj, pmask = synth.make_object()
    scenes = synth.kitti_like(12, 3, 375, 1242, torch.Generator().manual_seed(18)).to(dev)
    out = []
    for use_roi in (False, True):
        atk = Phy_obj_atk_l0(model, obj.to(dev), pmask.to(dev), adam_lr=0.5, steps=3, mask_wt=0.06, l0_thresh=0.1,
                             dist_range=list(np.arange(5, 10, 0.2)))
        atk.use_roi = use_roi
        atk.trace, atk.grad_trace = [], []
        random.seed(13)
        np.random.seed(13)
        adv, ben, m, patch = atk(scenes, 12)
        out.append((atk.trace, atk.grad_trace, atk.pattern_pos_tensor.detach().cpu(), atk.pattern_neg_tensor.detach().cpu(),
                    patch.cpu(), m.cpu()))
    (t0, g0, pp0, pn0, pa0, m0), (t1, g1, pp1, pn1, pa1, m1) = out
    assert len(t0) == len(t1) >= 3
    assert torch.equal(m0, m1)
    for i, (a, b) in enumerate(zip(t0, t1)):
        # L0 count (texels on the 1/255 threshold after Adam steps from gradients that agree to 1e-6), mask weight
        assert abs(a[0] - b[0]) <= max(5, 2e-3 * a[0]) and a[1] == b[1], (i, a, b)
        tol = 2e-6 if i == 0 else 1e-5      # Adam(lr = 0.5) turns the sign of a ~0 gradient into a step of 0.5 on that texel
        assert abs(a[2] - b[2]) <= tol * abs(a[2]) and abs(a[3] - b[3]) <= tol * abs(a[3]), (i, a, b)  # adversarial / mask cost
    per_iter = []
    for (gp0, gn0), (gp1, gn1) in zip(g0, g1):
        per_iter.append(max(float((x - y).double().norm() / x.double().norm()) for x, y in ((gp0, gp1), (gn0, gn1))))
    print("L0 attack, windows on vs off: pattern gradients rel-L2 per iteration %s" % ["%.3g" % v for v in per_iter])
    # iteration 0 starts from identical patterns: the same gradient up to the reduction order of the cost.  Adam(lr = 0.5)
    # then moves every texel by ~0.5 in the direction of its gradient's sign, so a texel whose gradient is ~0 takes the other
    # direction in one of the runs and the later iterations compare gradients at (slightly) different patterns
    assert per_iter[0] <= 2e-6, per_iter
    assert max(per_iter) <= 0.25, per_iter
    # The two runs are the same attack as FUNCTIONS (costs and L0 counts above, every iteration).  Texel by texel the patterns
    # part ways: Adam's update lr * m / (sqrt(v) + eps) is scale-free, so where the gradient is ~0 (texels the sparse sampler
    # hardly touches) its last bits decide a step of up to lr = 0.5 -- reported, not gated
    for name, x, y in (("pattern+", pp0, pp1), ("pattern-", pn0, pn1), ("patch", pa0, pa1)):
        print("%s texels within 1e-4 with / without windows: %.4f" % (name, ((x - y).abs() <= 1e-4).float().mean().item()))
    assert float((pa0 - pa1).abs().max()) <= 1.0 and torch.isfinite(pa0).all() and torch.isfinite(pa1).all()


@pytest.mark.gpu
def test_windowed_encoder_head_gradient_equals_the_full_one_inside_the_window():
    """ops.encoder_head_eval: features identical to the separate nodes; d / d image identical inside the plan's image
    window for an upstream gradient that is dense on feature 1 and lives inside "r_f0" on feature 0, zero outside it."""
    from depthmodelhardening_amd import ops
    from depthmodelhardening_amd.my_utils import to_device_async
    dev = torch.device("cuda")
    model = _unet(dev, seed=4)
    enc = model.encoder
    pt, grid = _pose_grid()
    rng = np.random.RandomState(7)
    g = torch.Generator().manual_seed(11)
    B = 12
    with ops.frozen_weights():
        for trial in range(2):
            poses = [grid[i] for i in rng.choice(len(grid), B, replace=False)]
            if trial == 0:
                poses[:4] = [(5.0, 0), (5.0, 30), (9.8, -30), (5.0, -30)]
            plan = RoiPlan(pt.mask_boxes([p[0] for p in poses], [p[1] for p in poses], (H, W)), H, W)
            tab = to_device_async(plan.table(), dev)
            x0 = torch.rand(B, 3, H, W, generator=g).to(dev)
            g_f1 = torch.randn(B, 64, H // 4, W // 4, generator=g).to(dev)
            g_f0 = torch.zeros(B, 64, H // 2, W // 2)
            (rh, rw), ro = plan.size["r_f0"], plan.org["r_f0"]
            for b in range(B):
                g_f0[b, :, ro[b, 0]:ro[b, 0] + rh, ro[b, 1]:ro[b, 1] + rw] = torch.randn(64, rh, rw, generator=g)
            g_f0 = g_f0.to(dev)
            res = []
            for roi in (None, (plan, tab)):
                x = x0.clone().requires_grad_(True)
                feats = enc(x, roi=roi)
                (gx,) = torch.autograd.grad([feats[0], feats[1]], x, [g_f0, g_f1])
                res.append((feats[0].detach(), feats[1].detach(), gx))
            assert plan.head_windowed
            assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
            (hd, wd), od = plan.size["d"], plan.org["d"]
            inside = torch.zeros(B, 1, H, W, device=dev)
            for b in range(B):
                inside[b, :, od[b, 0]:od[b, 0] + hd, od[b, 1]:od[b, 1] + wd] = 1
            full, win = res[0][2], res[1][2]
            assert float((win * (1 - inside)).abs().max()) == 0.0
            err = float(((win - full) * inside).abs().max()) / float((full * inside).abs().max())
            print("windowed encoder head vs full, inside the image window: max err / max |g| = %.3g" % err)
            # not bitwise: K12's window form adds its 64 gradient channels as four partial sums of 16 (one per wave), the
            # whole-frame form in one chain -- a 2,352-product sum rounds ~sqrt(2352) * 2^-24 = 3e-6 of its terms either way
            assert err <= 2e-6


@pytest.mark.gpu
def test_incremental_encoder_head_equals_the_full_one():
    """ops.encoder_head_incremental on pasted frames + their clean frames: feature 1 and the "hz" window of feature 0 are
    bitwise those of the whole-frame head, d / d image equals the whole-frame gradient inside the image window, the cached
    clean feature is intact again after restore(), and a step whose frames differ from the clean ones OUTSIDE the box is
    what the construction excludes (the result then differs: the test shows the check bites)."""
    from depthmodelhardening_amd import ops
    from depthmodelhardening_amd.my_utils import to_device_async
    from oracle import synth
    dev = torch.device("cuda")
    model = _unet(dev, seed=6)
    enc = model.encoder
    pt, grid = _pose_grid()
    obj, pmask = synth.make_object()
    obj, pmask = obj.to(dev), pmask.to(dev)
    scenes = synth.kitti_like(12, 3, 375, 1242, torch.Generator().manual_seed(15)).to(dev)
    rng = np.random.RandomState(9)
    g = torch.Generator().manual_seed(12)
    B = 12
    with ops.frozen_weights():
        clean, _ = ops.eot_paste(scenes, obj, torch.zeros_like(pmask), to_device_async(pt.coeffs_for([7.0] * B, [0] * B), dev),
                                 pt.l_pad, pt.t_pad, (H, W))
        for trial in range(3):
            poses = [grid[i] for i in rng.choice(len(grid), B, replace=False)]
            if trial == 0:
                poses[:4] = [(5.0, 0), (5.0, 30), (9.8, -30), (5.0, -30)]
            z0, al = [p[0] for p in poses], [p[1] for p in poses]
            plan_full = RoiPlan(pt.mask_boxes(z0, al, (H, W)), H, W)
            plan = RoiPlan(pt.mask_boxes(z0, al, (H, W)), H, W)
            assert plan.head_incremental_ok
            tab = to_device_async(plan.table(), dev)
            x0, m = ops.eot_paste(scenes, obj, pmask, to_device_async(pt.coeffs_for(z0, al), dev), pt.l_pad, pt.t_pad, (H, W))
            assert float(((x0 - clean) * (m == 0)).abs().max()) == 0.0       # the premise: equal outside the mask
            g_f1 = torch.zeros(B, 64, H // 4, W // 4)           # the decoder's skip gradient: lives inside "r_f1"
            (rh, rw), ro = plan.size["r_f1"], plan.org["r_f1"]
            for b in range(B):
                g_f1[b, :, ro[b, 0]:ro[b, 0] + rh, ro[b, 1]:ro[b, 1] + rw] = torch.randn(64, rh, rw, generator=g)
            g_f1 = g_f1.to(dev)
            g_f2 = torch.randn(B, 128, H // 8, W // 8, generator=g).to(dev)      # dense: it comes down from layer3
            hz, oz = plan.size["hz"], plan.org["hz"]
            g_f0c = torch.randn(B, 64, hz[0], hz[1], generator=g).to(dev)
            g_f0 = torch.zeros(B, 64, H // 2, W // 2, device=dev)
            for b in range(B):
                g_f0[b, :, oz[b, 0]:oz[b, 0] + hz[0], oz[b, 1]:oz[b, 1] + hz[1]] = g_f0c[b]
            # whole-frame reference (the plain nodes)
            x = x0.clone().requires_grad_(True)
            ref = enc(x)
            (gx_ref,) = torch.autograd.grad([ref[0], ref[1], ref[2]], x, [g_f0, g_f1, g_f2])
            # incremental
            x = x0.clone().requires_grad_(True)
            feats = enc(x, roi=(plan, tab), clean=clean)
            assert plan.f0_compact and plan.head_windowed and tuple(feats[0].shape[2:]) == hz
            assert torch.equal(feats[1], ref[1])
            for b in range(B):
                assert torch.equal(feats[0][b], ref[0][b, :, oz[b, 0]:oz[b, 0] + hz[0], oz[b, 1]:oz[b, 1] + hz[1]])
            for k in (2, 3, 4):
                assert torch.equal(feats[k], ref[k])
            (gx,) = torch.autograd.grad([feats[0], feats[1], feats[2]], x, [g_f0c, g_f1, g_f2])
            (hd, wd), od = plan.size["d"], plan.org["d"]
            inside = torch.zeros(B, 1, H, W, device=dev)
            for b in range(B):
                inside[b, :, od[b, 0]:od[b, 0] + hd, od[b, 1]:od[b, 1] + wd] = 1
            assert float((gx * (1 - inside)).abs().max()) == 0.0
            err = float(((gx - gx_ref) * inside).abs().max()) / float((gx_ref * inside).abs().max())
            print("incremental encoder head vs full, inside the image window: max err / max |g| = %.3g" % err)
            assert err <= 2e-6        # K12's window form: four partial channel sums, see the test above
            del plan_full
        cache = ops.frozen_memo(("clean_head", id(enc), clean.data_ptr(), clean._version, True), lambda: None)
        assert cache is not None and cache.dirty is not None and not torch.equal(cache.work, cache.pristine)
        assert cache.dirty2 is not None and not torch.equal(cache.work2, cache.pristine2)
        cache.restore()
        assert torch.equal(cache.work, cache.pristine) and torch.equal(cache.work2, cache.pristine2)
        # two steps' graphs cannot be alive across each other's forward: the cached features are re-written in place
        xa = x0.clone().requires_grad_(True)
        fa = enc(xa, roi=(plan, tab), clean=clean)
        enc(x0.clone().requires_grad_(True), roi=(plan, tab), clean=clean)
        with pytest.raises(RuntimeError, match="later forward"):
            torch.autograd.grad(fa[1].sum(), xa)
        # the check bites: frames that differ from the clean ones outside the box give a different feature 1
        x_bad = x0.clone()
        x_bad[:, :, :8, :8] += 0.25
        feats_bad = enc(x_bad.requires_grad_(True), roi=(plan, tab), clean=clean)
        assert not torch.equal(feats_bad[1], enc(x_bad)[1])


@pytest.mark.gpu
def test_windowed_attack_step_with_clean_frames_equals_the_full_frame_step():
    """The attack step with every K19 shortcut on (windows + incremental head) against the whole-frame step."""
    from depthmodelhardening_amd import ops
    from depthmodelhardening_amd.my_utils import to_device_async
    from oracle import synth
    dev = torch.device("cuda")
    model = _unet(dev, seed=3)
    pt, grid = _pose_grid()
    obj, pmask = synth.make_object()
    obj, pmask = obj.to(dev), pmask.to(dev)
    scenes = synth.kitti_like(12, 3, 375, 1242, torch.Generator().manual_seed(25)).to(dev)
    rng = np.random.RandomState(14)
    with ops.frozen_weights():
        clean, _ = ops.eot_paste(scenes, obj, torch.zeros_like(pmask), to_device_async(pt.coeffs_for([7.0] * 12, [0] * 12), dev),
                                 pt.l_pad, pt.t_pad, (H, W))
        for trial in range(3):
            poses = [grid[i] for i in rng.choice(len(grid), 12, replace=False)]
            z0, al = [p[0] for p in poses], [p[1] for p in poses]
            coeffs = to_device_async(pt.coeffs_for(z0, al), dev)
            res = []
            for depth in (None, 2, 4):
                plan = tab = None
                if depth is not None:
                    plan = RoiPlan(pt.mask_boxes(z0, al, (H, W)), H, W, depth=depth)
                    tab = to_device_async(plan.table(), dev)
                patch = obj.clone().requires_grad_(True)
                adv, m = ops.eot_paste(scenes, patch, pmask, coeffs, pt.l_pad, pt.t_pad, (H, W))
                cost = -(model.masked_sq_mean(adv, m, plan, tab, clean) if plan is not None else ops.masked_sq_mean(model(adv), m))
                (grad,) = torch.autograd.grad(cost, patch)
                res.append((cost.detach().double().cpu(), grad.detach().double().cpu()))
                assert plan is None or (plan.head_windowed and plan.f0_compact)
            c0, g0 = res[0]
            for c1, g1 in res[1:]:
                assert abs(float(c1 - c0)) <= 1e-6 * abs(float(c0))
                assert float((g1 - g0).norm() / g0.norm()) <= 1e-6
                assert float((g1 - g0).abs().max()) <= 2e-5 * float(g0.abs().max()), trial


@pytest.mark.gpu
def test_attack_step_replays_bit_identically_from_a_hip_graph():
    """the ops launch on the current stream, never synchronise and allocate through PyTorch's caching
    allocator only, so a whole attack step -- K3 paste, the U-Net with its K19 windows, the cost, autograd's backward, K3's
    adjoint, K4 -- can be captured once with torch.cuda.graph and replayed: the replay's patch equals the eager step's bit for
    bit, and a second replay from the same input gives the same bits again."""
    from depthmodelhardening_amd import ops
    from depthmodelhardening_amd.my_utils import to_device_async
    from oracle import synth
    dev = torch.device("cuda")
    model = _unet(dev, seed=8)
    pt, grid = _pose_grid()
    obj, pmask = synth.make_object()
    obj, pmask = obj.to(dev), pmask.to(dev)
    scenes = synth.kitti_like(12, 3, 375, 1242, torch.Generator().manual_seed(35)).to(dev)
    rng = np.random.RandomState(24)
    poses = [grid[i] for i in rng.choice(len(grid), 12, replace=False)]
    z0, al = [p[0] for p in poses], [p[1] for p in poses]
    coeffs = to_device_async(pt.coeffs_for(z0, al), dev)
    plan = RoiPlan(pt.mask_boxes(z0, al, (H, W)), H, W, depth=ops.ROI_DEPTH)
    tab = to_device_async(plan.table(), dev)
    patch_in = obj.clone()
    patch_out = torch.empty_like(obj)

    def step():
        p = patch_in.detach().requires_grad_(True)
        adv, m = ops.eot_paste(scenes, p, pmask, coeffs, pt.l_pad, pt.t_pad, (H, W))
        cost = -model.masked_sq_mean(adv, m, plan, tab, clean)
        (grad,) = torch.autograd.grad(cost, p)
        ops.pgd_linf_step(p, obj, grad, 0.02, 0.1, out=patch_out)

    with ops.frozen_weights():
        clean, _ = ops.eot_paste(scenes, obj, torch.zeros_like(pmask), coeffs, pt.l_pad, pt.t_pad, (H, W))
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):       # warm-up on the side stream: filter transforms, the clean-frame cache, LDS limits
            step()
            step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        eager = patch_out.clone()
        assert not torch.equal(eager, obj)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        outs = []
        for _ in range(2):
            patch_out.zero_()
            graph.replay()
            torch.cuda.synchronize()
            outs.append(patch_out.clone())
        assert torch.equal(outs[0], eager) and torch.equal(outs[1], eager)
        # the graph reads its inputs in place: a new input patch gives the step from THAT patch
        patch_in.copy_(eager)
        graph.replay()
        torch.cuda.synchronize()
        assert not torch.equal(patch_out, eager) and float((patch_out - obj).abs().max()) <= 0.1 + 1e-6
