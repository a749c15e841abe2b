"""HIP-vs-oracle parity for every kernel family, through the C ABI (libdmh_hip.so).

Tolerance: north_star asks for 1e-4 relative fp32.  Scalars (losses) are held to 2e-5;
per-element tensors to 1e-4 of the tensor's scale, with a tiny outlier allowance only where the
function is non-smooth in its inputs (floor() of the bilinear sampler, min/argmin ties).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from tests.util import GradPool, assert_close_frac, np_t, rel_l2, to_dev  # noqa: E402


def _mods():
    from depthmodelhardening_amd import _native as N, ops
    from oracle import attack_ref, loss_ref, synth, tv082
    return N, ops, loss_ref, attack_ref, synth, tv082


# Per-pixel SSIM values carry fp32 conditioning noise: sigma = E[x^2] - mu^2 cancels ~0.25 - 0.25 down to
# ~1e-3, so one ulp of the window sums moves (1 - SSIM)/2 by ~1e-5.  Scalars (means over >= 6k pixels) are
# held to 2e-5 relative; per-pixel maps to 5e-5 absolute.
PIX_ATOL = 5e-5


def _near_tie_exclusion(loss_ref, inputs, outputs, noise, scale, variant, B, H, W, frame_ids=(0, "s")):
    """Pixels whose identity/reprojection decision is within fp32 noise in the ORACLE, dilated to every
    gradient element they can reach (3x3 SSIM window, then the bilinear up-sampling footprint)."""
    tgt = inputs[("color", 0, 0)]
    ident = torch.cat([loss_ref.compute_reprojection_loss(inputs[("color", f, 0)], tgt) for f in frame_ids[1:]], 1)
    reproj = torch.cat([loss_ref.compute_reprojection_loss(outputs[("color", f, scale)], tgt)
                        for f in frame_ids[1:]], 1).detach()
    if variant == "dh":
        ident = ident.min(1, keepdim=True)[0]
    if noise is not None:
        ident = ident + noise[scale]
    gap = (ident.min(1)[0] - reproj.min(1)[0]).abs()
    tie = (gap < 2 * PIX_ATOL).float().view(B, 1, H, W)
    full = F.max_pool2d(tie, 3, 1, 1)
    f = 2 ** scale
    low = full if f == 1 else F.max_pool2d(full, 3 * f, f, f)
    return tie.view(B, H, W) > 0, low > 0


def _oracle_loss(loss_ref, inputs, disps, noise, variant, frame_ids=(0, "s")):
    outputs, leaves = {}, []
    for s, d in enumerate(disps):
        d = d.clone().requires_grad_(True)
        leaves.append(d)
        outputs[("disp", s)] = d
    loss_ref.generate_images_pred(inputs, outputs, frame_ids=frame_ids)
    losses, maps = loss_ref.compute_losses(inputs, outputs, frame_ids=frame_ids, noise=noise, variant=variant)
    losses["loss"].backward()
    return losses, maps, outputs, [l.grad for l in leaves]


def _pool_seeds(B, H, W, seed):
    """Seeds to pool so that the gradient bound is decided by >= ~1.2e5 pixels, not by a handful of flips -- and no more than
    that: every seed is one float32 and one float64 oracle pass on the CPU (the suite's time budget, profiles/
    r06_gpu_suite_durations.txt)."""
    n = int(min(24, max(1, -(-120000 // (B * H * W)))))
    return [seed + 1000 * i for i in range(n)]


@pytest.mark.parametrize("variant", ["md2", "dh"])
@pytest.mark.parametrize("shape", [(32, 32, 96, 21), (2, 192, 640, 22), (24, 48, 80, 5), (12, 64, 200, 9)])
@pytest.mark.parametrize("with_noise", [False, True])
def test_photo_smooth_loss_vs_oracle(variant, shape, with_noise):
    # (the three small image sizes run at batch 32 / 24 / 12: the pooled pixel count is what decides the gradient bound, and a
    # seed costs two oracle passes whose time at these sizes is all per-call overhead; batch 2 at such sizes: the goldens)
    if not with_noise and shape[1:3] != (192, 640):
        pytest.skip("the noise-free form is covered at the config-1 shape (and by the conditioning tests and the goldens); with "
                    "the tie-break noise at all four (suite time)")
    N, ops, loss_ref, _, synth, _ = _mods()
    B, H, W, seed0 = shape
    pool = GradPool(count_floor=2.0 / (H * W) if variant == "dh" else 0.0)
    for seed in _pool_seeds(B, H, W, seed0):
        inputs, disps = synth.make_loss_case(B, H, W, seed)
        noise = None
        if with_noise:
            g = torch.Generator().manual_seed(seed + 100)
            noise = {s: torch.randn(B, 1, H, W, generator=g) * 0.00001 for s in range(4)}
        losses, maps, outputs, grads = _oracle_loss(loss_ref, inputs, disps, noise, variant)
        in64, disps64 = synth.make_loss_case(B, H, W, seed, dtype=torch.float64)
        noise64 = None if noise is None else {s: z.double() for s, z in noise.items()}
        losses64, _, _, grads64 = _oracle_loss(loss_ref, in64, disps64, noise64, variant)

        d_in = to_dev(inputs)
        d_disps = [d.cuda().requires_grad_(True) for d in disps]
        out = ops.photometric_smooth_loss(
            d_in[("color", 0, 0)], [d_in[("color", "s", 0)]], [d_in["stereo_T"]], d_in[("K", 0)], d_in[("inv_K", 0)],
            d_disps, [d_in[("color", 0, s)] for s in range(4)], variant=variant,
            noise=None if noise is None else [noise[s].cuda() for s in range(4)], want_to_opt=True)
        fin = out.fin
        fin[N.FIN_LOSS].backward()
        torch.cuda.synchronize()
        f = fin.detach().cpu()
        # md2's mean(min(.)) is continuous in the inputs: 2e-5 against the fp32 oracle.  dh's masked-sum / mask-count jumps
        # by (value - mean)/count whenever an fp32 near-tie flips the argmin -- in the fp32 oracle as much as in the kernel
        # -- so it is anchored on the fp64 oracle like the gradients: HIP may be 1.5x as far from fp64 as the fp32 oracle
        # is, plus two flips.

        def scalar_ok(got, key):
            if variant == "md2":
                return abs(got - losses[key].item()) <= 2e-5 * abs(losses[key].item())
            ref64 = losses64[key].item()
            return abs(got - ref64) <= 1.5 * abs(losses[key].item() - ref64) + max(2e-5, 2.0 / (B * H * W)) * abs(ref64)
        assert scalar_ok(f[N.FIN_LOSS].item(), "loss"), (f[N.FIN_LOSS].item(), losses["loss"].item(), losses64["loss"].item())
        for s in range(4):
            key = "loss/%d" % s
            assert scalar_ok(f[N.FIN_LOSS_S + s].item(), key), (s, f[N.FIN_LOSS_S + s].item(), losses[key].item(),
                                                               losses64[key].item())
            sel_ref = outputs["identity_selection/%d" % s].reshape(B, H, W)
            sel = out.sel[s].cpu()
            if variant == "dh":
                sel = 1.0 - (sel > 0).float()
            tie, excl = _near_tie_exclusion(loss_ref, inputs, outputs, noise, s, variant, B, H, W)
            assert tie.float().mean().item() < 0.01
            assert ((sel != sel_ref) & ~tie).sum().item() == 0, "selection mask differs away from fp32 ties"
            ref_map = maps[s].reshape(B, H, W)
            if variant == "dh":   # masked map: a flipped tie moves the value by the whole loss, compare off-tie only
                assert_close_frac(out.to_opt[s].cpu()[~tie], ref_map[~tie], rtol=1e-4, atol=PIX_ATOL,
                                  name="to_opt[%d]" % s)
            else:
                assert_close_frac(out.to_opt[s], ref_map, rtol=1e-4, atol=PIX_ATOL, name="to_opt[%d]" % s)
            # Gradients: anchored on the SAME oracle in float64, over ALL elements (no tie exclusion, no trimming), pooled
            # over the seeds.  The fp32 oracle's own distance to fp64 is the conditioning noise of this loss (floor() of
            # the sampler and argmin flips, see tests/test_gpu_parity_anchor.py); HIP must stay within 1.5x of it.
            pool.add(s, d_disps[s].grad, grads[s], grads64[s])
    pool.check("%s %s noise=%s" % (variant, shape, with_noise))


@pytest.mark.parametrize("name", ["loss_md2_cfg1", "loss_md2_hd"])
def test_photo_loss_golden_cfg1(golden, name):
    """HIP path against the reference's own numbers (tests/golden): BASELINE config-1 shape 2 x 192 x 640, and ``hd`` =
    2 x 320 x 1024, the resolution of configs 2-5 -- K1's 62 / 60-column strips and 16 / 32-row strips tile it differently,
    and the fp32 coordinate noise the tolerance model rests on grows with x (the hd fixture keeps gradients of its noise-free
    run only)."""
    N, ops, loss_ref, _, synth, _ = _mods()
    g = golden(name)
    B, H, W, seed = [int(v) for v in g["shape"]]
    inputs, disps = synth.make_loss_case(B, H, W, seed)
    gen = torch.Generator().manual_seed(seed + 100)
    noise = [torch.randn(B, 1, H, W, generator=gen) * 0.00001 for _ in range(4)]
    d_in = to_dev(inputs)
    in64, disps64 = synth.make_loss_case(B, H, W, seed, dtype=torch.float64)
    grads64 = {"nonoise": _oracle_loss(loss_ref, in64, disps64, None, "md2")[3],
               "noise": _oracle_loss(loss_ref, in64, disps64, {s: z.double() for s, z in enumerate(noise)}, "md2")[3]}
    for tag, nz in (("nonoise", None), ("noise", [z.cuda() for z in noise])):
        d_disps = [d.cuda().requires_grad_(True) for d in disps]
        out = ops.photometric_smooth_loss(
            d_in[("color", 0, 0)], [d_in[("color", "s", 0)]], [d_in["stereo_T"]], d_in[("K", 0)],
            d_in[("inv_K", 0)], d_disps, [d_in[("color", 0, s)] for s in range(4)], noise=nz)
        out.fin[N.FIN_LOSS].backward()
        f = out.fin.detach().cpu()
        assert abs(f[N.FIN_LOSS].item() - float(g[tag + "_loss"])) <= 2e-5 * abs(float(g[tag + "_loss"]))
        for s in range(4):
            ref = float(g["%s_loss_%d" % (tag, s)])
            assert abs(f[N.FIN_LOSS_S + s].item() - ref) <= 2e-5 * abs(ref)
            if "%s_identity_selection_%d" % (tag, s) not in g.files:
                continue
            sel = np.unpackbits(g["%s_identity_selection_%d" % (tag, s)])[:B * H * W].reshape(B, H, W)
            assert (out.sel[s].cpu().numpy() != sel).mean() <= 5e-4   # fp32 near-ties only
            # gradients: the golden is an fp32 run of the reference; both it and the HIP result are measured against
            # the fp64 oracle and HIP may not be further away than 1.5x the reference's own fp32 run
            key = "%s_grad_disp_%d" % (tag, s)
            g64 = grads64[tag][s]
            if key in g.files:
                ref_g, got_g, a64 = np_t(g[key]).double(), d_disps[s].grad.double().cpu(), g64
            else:
                ref_g, got_g, a64 = np_t(g[key + "_sub3"]).double(), d_disps[s].grad[:, :, ::3, ::3].double().cpu(), \
                    g64[:, :, ::3, ::3]
                got = d_disps[s].grad.double().abs().sum((1, 2, 3)).cpu()
                torch.testing.assert_close(got, np_t(g[key + "_abssum"]), rtol=5e-3, atol=0)
            e_h, e_r = rel_l2(got_g, a64), rel_l2(ref_g, a64)
            tol_el = 1e-4 * a64.abs().max().item() + 1e-4 * a64.abs()
            n_h = int(((got_g - a64).abs() > tol_el).sum())
            n_r = int(((ref_g - a64).abs() > tol_el).sum())
            assert n_h <= 1.5 * n_r + 1e-3 * a64.numel(), (key, n_h, n_r)
            # systematic accuracy, absolute: on the elements HIP has within tolerance (all but the n_h counted above) it
            # agrees with float64 to 5e-5 in rel-L2 (measured 4e-6 at scale 0 ... 2e-5 at scale 3)
            core = (got_g - a64).abs() <= tol_el
            e_core = float((got_g - a64)[core].norm() / a64[core].norm())
            print("%s %s: vs fp64 rel-L2 hip %.3g reference %.3g | beyond tolerance hip %d reference %d of %d | HIP on its "
                  "in-tolerance elements %.3g" % (name, key, e_h, e_r, n_h, n_r, a64.numel(), e_core))
            assert e_core <= 5e-5, (key, e_core)
            if name == "loss_md2_cfg1":
                assert e_h <= 1.5 * e_r + 1e-6, (key, e_h, e_r)
            # at 320 x 1024 ONE pixel decides the untrimmed rel-L2 of a single case: (b 1, y 294, x 567) of this fixture has
            # its sample 4.6e-6 px from a texel boundary in float64 (an fp32 coordinate at x = 567 resolves 6e-5, HIP's
            # pixel + delta form ~2e-6) and carries 99 % of HIP's squared error, while the reference's fp32 run flips 94
            # other elements (tools/diag_grad_outliers.py 2 320 1024 23); the pooled form of the rel-L2 bound is
            # test_gradients_within_fp32_conditioning_of_the_fp64_oracle at this shape (three seeds, all elements)


def _general_pose(B, seed):
    """A different rigid transform per sample: rotations of up to ~1 degree about all three axes + a translation."""
    g = torch.Generator().manual_seed(seed)
    ang = (torch.rand(B, 3, generator=g, dtype=torch.float64) - 0.5) * 0.03
    T = torch.eye(4, dtype=torch.float64).repeat(B, 1, 1)
    for b in range(B):
        cx, cy, cz = torch.cos(ang[b])
        sx, sy, sz = torch.sin(ang[b])
        Rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=torch.float64)
        Ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=torch.float64)
        Rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=torch.float64)
        T[b, :3, :3] = Rz @ Ry @ Rx
    T[:, 0, 3], T[:, 1, 3], T[:, 2, 3] = 0.05, 0.01, -0.02
    T[:, :3, 3] += (torch.rand(B, 3, generator=g, dtype=torch.float64) - 0.5) * 0.02
    return T


def test_photo_loss_two_frames_and_options():
    """Two source frames (min over frames) with a general pose (rotation + translation, different per sample) for the
    second one; then --no_ssim, --disable_automasking.  Gradients pooled over seeds against the fp64 oracle."""
    N, ops, loss_ref, _, synth, _ = _mods()
    B, H, W = 6, 40, 136                # (6 images x 4 seeds: the pixel count of the 2 x 12 it replaces, a third of the oracle calls)
    fids = (0, -1, "s")
    pool = GradPool()
    for seed in (77, 177, 277, 377):
        inputs, disps = synth.make_loss_case(B, H, W, seed)
        g = torch.Generator().manual_seed(seed + 1)
        inputs[("color", -1, 0)] = (0.8 * torch.roll(inputs[("color", 0, 0)], -2, 3) +
                                    0.2 * synth.kitti_like(B, 3, H, W, g))
        T2_64 = _general_pose(B, seed + 2)
        T2 = T2_64.float()
        d_in = to_dev(inputs)

        def oracle(dtype):
            ins = {k: v.to(dtype) for k, v in inputs.items()}
            outs, leaves = {("cam_T_cam", 0, -1): T2.to(dtype)}, []   # the fp32-rounded pose on both sides
            for s, d in enumerate(disps):
                d = d.detach().clone().to(dtype).requires_grad_(True)
                leaves.append(d)
                outs[("disp", s)] = d
            loss_ref.generate_images_pred(ins, outs, frame_ids=fids)
            ls, mp = loss_ref.compute_losses(ins, outs, frame_ids=fids, noise=None, variant="md2")
            ls["loss"].backward()
            return ls, mp, leaves
        losses, maps, leaves = oracle(torch.float32)
        _, _, leaves64 = oracle(torch.float64)
        d_disps = [d.cuda().requires_grad_(True) for d in disps]
        out = ops.photometric_smooth_loss(d_in[("color", 0, 0)], [d_in[("color", -1, 0)], d_in[("color", "s", 0)]],
                                          [T2.cuda(), d_in["stereo_T"]], d_in[("K", 0)], d_in[("inv_K", 0)], d_disps,
                                          [d_in[("color", 0, s)] for s in range(4)], noise=None, want_to_opt=True)
        out.fin[N.FIN_LOSS].backward()
        ref = losses["loss"].item()
        assert abs(out.fin[N.FIN_LOSS].item() - ref) <= 2e-5 * abs(ref)
        for s in range(4):
            assert_close_frac(out.to_opt[s], maps[s], rtol=1e-4, atol=PIX_ATOL, name="to_opt2[%d]" % s)
            pool.add(s, d_disps[s].grad, leaves[s].grad, leaves64[s].grad)
    pool.check("two frames, general pose")
    inputs, disps = synth.make_loss_case(B, H, W, 77)
    d_in = to_dev(inputs)
    # no_ssim + no automask: plain mean L1
    out2 = ops.photometric_smooth_loss(d_in[("color", 0, 0)], [d_in[("color", "s", 0)]], [d_in["stereo_T"]],
                                       d_in[("K", 0)], d_in[("inv_K", 0)], [d.cuda() for d in disps],
                                       [d_in[("color", 0, s)] for s in range(4)], noise=None, automask=False,
                                       no_ssim=True, want_to_opt=True)
    outputs = {("disp", s): disps[s] for s in range(4)}
    loss_ref.generate_images_pred(inputs, outputs)
    for s in range(4):
        l1 = loss_ref.compute_reprojection_loss(outputs[("color", "s", s)], inputs[("color", 0, 0)], no_ssim=True)
        assert_close_frac(out2.to_opt[s], l1[:, 0], rtol=1e-4, atol=5e-6, max_bad_frac=1e-4, name="l1[%d]" % s)
        assert (out2.sel[s] == 1).all()


@pytest.mark.parametrize("variant", ["md2", "dh"])
def test_pose_gradient_vs_fp64_oracle(variant):
    """d loss / d cam_T_cam (the pose of a monocular source frame, MD2/trainer.py:487-519 -> layers.py:182-198) and
    d loss / d stereo_T from K1's backward == autograd of the oracle: pooled over seeds, the HIP result is as close to the
    float64 oracle as the float32 oracle is (x1.5 + 1e-4 of the gradient's norm); disparity gradients unchanged by the
    pose path; bitwise reproducible."""
    N, ops, loss_ref, _, synth, _ = _mods()
    B, H, W = 2, 40, 136
    fids = (0, -1, "s")
    e_h = e_o = n64 = 0.0
    for seed in (31, 131, 231, 331, 431, 531):
        inputs, disps = synth.make_loss_case(B, H, W, seed)
        g = torch.Generator().manual_seed(seed + 1)
        inputs[("color", -1, 0)] = (0.8 * torch.roll(inputs[("color", 0, 0)], -2, 3) +
                                    0.2 * synth.kitti_like(B, 3, H, W, g))
        T2 = _general_pose(B, seed + 2).float()
        d_in = to_dev(inputs)

        def oracle(dtype):
            ins = {k: v.to(dtype) for k, v in inputs.items()}
            Ta = T2.detach().clone().to(dtype).requires_grad_(True)      # (.to() alone returns T2 itself for float32)
            Ts = ins["stereo_T"].detach().clone().requires_grad_(True)
            ins["stereo_T"] = Ts
            outs = {("cam_T_cam", 0, -1): Ta}
            for s, d in enumerate(disps):
                outs[("disp", s)] = d.detach().clone().to(dtype)
            loss_ref.generate_images_pred(ins, outs, frame_ids=fids)
            ls, _ = loss_ref.compute_losses(ins, outs, frame_ids=fids, noise=None, variant=variant)
            ls["loss"].backward()
            return Ta.grad, Ts.grad
        g32, g64 = oracle(torch.float32), oracle(torch.float64)

        def hip(pose):
            Ta = T2.detach().cuda().requires_grad_(pose)
            Ts = d_in["stereo_T"].detach().clone().requires_grad_(pose)
            dd = [d.cuda().requires_grad_(True) for d in disps]
            out = ops.photometric_smooth_loss(d_in[("color", 0, 0)], [d_in[("color", -1, 0)], d_in[("color", "s", 0)]],
                                              [Ta, Ts], d_in[("K", 0)], d_in[("inv_K", 0)], dd,
                                              [d_in[("color", 0, s)] for s in range(4)], noise=None, variant=variant)
            out.fin[N.FIN_LOSS].backward()
            return Ta.grad, Ts.grad, [d.grad for d in dd]
        ga, gs, gd = hip(True)
        ga2, gs2, _ = hip(True)
        assert torch.equal(ga, ga2) and torch.equal(gs, gs2)
        na, ns, gd0 = hip(False)
        assert na is None and ns is None
        for a_, b_ in zip(gd, gd0):        # the pose variant of the kernel computes the same disparity gradients
            torch.testing.assert_close(a_, b_, rtol=1e-4, atol=1e-6 * float(b_.abs().max()))    # (another instruction order)
        for got, r32, r64 in ((ga, g32[0], g64[0]), (gs, g32[1], g64[1])):
            assert float(got[:, 3].abs().max()) == 0.0            # (K T)[:3,:] does not depend on T's last row
            e_h += float((got.cpu().double() - r64).pow(2).sum())
            e_o += float((r32.double() - r64).pow(2).sum())
            n64 += float(r64.pow(2).sum())
    e_h, e_o, n64 = e_h ** 0.5, e_o ** 0.5, n64 ** 0.5
    assert n64 > 0 and e_h <= 1.5 * e_o + 1e-4 * n64, (e_h / n64, e_o / n64)


def test_philox_noise_is_tiny_and_seeded():
    N, ops, loss_ref, _, synth, _ = _mods()
    B, H, W = 2, 32, 96
    inputs, disps = synth.make_loss_case(B, H, W, 3)
    d_in = to_dev(inputs)

    def run():
        return ops.photometric_smooth_loss(d_in[("color", 0, 0)], [d_in[("color", "s", 0)]], [d_in["stereo_T"]],
                                           d_in[("K", 0)], d_in[("inv_K", 0)], [d.cuda() for d in disps],
                                           [d_in[("color", 0, s)] for s in range(4)], noise="philox",
                                           want_to_opt=True)
    base = ops.photometric_smooth_loss(d_in[("color", 0, 0)], [d_in[("color", "s", 0)]], [d_in["stereo_T"]],
                                       d_in[("K", 0)], d_in[("inv_K", 0)], [d.cuda() for d in disps],
                                       [d_in[("color", 0, s)] for s in range(4)], noise=None, want_to_opt=True)
    torch.manual_seed(123)
    a = run()
    torch.manual_seed(123)
    b = run()
    assert torch.equal(a.fin, b.fin) and torch.equal(a.to_opt[0], b.to_opt[0])
    c = run()  # next draw differs
    assert not torch.equal(a.to_opt[0], c.to_opt[0])
    # the tie-break term is N(0,1)*1e-5 on the identity branch only
    d = (a.to_opt[0] - base.to_opt[0])
    ident_px = (a.sel[0] == 0) & (base.sel[0] == 0)
    assert ident_px.any()
    z = d[ident_px] / 1e-5
    assert abs(z.mean().item()) < 0.2 and 0.7 < z.std().item() < 1.3 and z.abs().max().item() < 6
    assert abs(a.fin[N.FIN_LOSS].item() - base.fin[N.FIN_LOSS].item()) < 1e-5 * abs(base.fin[N.FIN_LOSS].item())


@pytest.mark.parametrize("scale", [0, 2])
def test_warp_view_vs_oracle(scale):
    N, ops, loss_ref, _, synth, _ = _mods()
    B, H, W = 2, 48, 112
    inputs, disps = synth.make_loss_case(B, H, W, 13)
    d = disps[scale].clone().requires_grad_(True)
    depth, grid, warped = loss_ref.warp_view(d, inputs[("color", "s", 0)], inputs[("K", 0)], inputs[("inv_K", 0)],
                                             inputs["stereo_T"], H, W)
    gc = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(1))
    gd = torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(2)) * 1e-3
    ((warped * gc).sum() + (depth * gd).sum()).backward()
    dd = disps[scale].cuda().requires_grad_(True)
    dp, smp, col = ops.warp_view(inputs[("color", "s", 0)].cuda(), dd, inputs[("K", 0)].cuda(),
                                 inputs[("inv_K", 0)].cuda(), inputs["stereo_T"].cuda(), H, W)
    ((col * gc.cuda()).sum() + (dp * gd.cuda()).sum()).backward()
    assert_close_frac(dp, depth, rtol=1e-5, atol=0, name="depth")
    assert_close_frac(smp, grid, rtol=1e-5, atol=2e-6, name="sample")
    assert_close_frac(col, warped, rtol=1e-4, atol=2e-5, max_bad_frac=1e-4, name="color")
    scale_g = d.grad.abs().max().item()
    assert_close_frac(dd.grad, d.grad, rtol=1e-3, atol=2e-4 * scale_g, max_bad_frac=2e-3, name="warp grad")


@pytest.mark.parametrize("sizes", [(2, 24, 40, 12, 20), (1, 320, 1024, 40, 128), (2, 30, 50, 15, 25), (1, 16, 16, 16, 16)])
def test_upsample_adjoint(sizes):
    N, ops, *_ = _mods()
    import ctypes as C
    B, H, W, Hs, Ws = sizes
    g_up = torch.rand(B, H, W, generator=torch.Generator().manual_seed(4)) - 0.5
    d = torch.zeros(B, 1, Hs, Ws, requires_grad=True)
    (F.interpolate(d, [H, W], mode="bilinear", align_corners=False)[:, 0] * g_up).sum().backward()
    out = torch.full((B, 1, Hs, Ws), 7.0, device="cuda")
    gu = g_up.cuda()
    N.check(N.lib().dmh_upsample_bilinear_adjoint(N.ptr(gu), N.ptr(out), B, H, W, Hs, Ws, 0, N.stream()))
    assert_close_frac(out, d.grad, rtol=1e-5, atol=1e-5, name="adjoint")
    N.check(N.lib().dmh_upsample_bilinear_adjoint(N.ptr(gu), N.ptr(out), B, H, W, Hs, Ws, 1, N.stream()))
    assert_close_frac(out, 2 * d.grad, rtol=1e-5, atol=2e-5, name="adjoint accumulate")


def _paste_case(attack_ref, synth, tv082, n, seed):
    obj, mask = synth.make_object()
    scenes = synth.kitti_like(n, 3, 375, 1242, torch.Generator().manual_seed(seed))
    pt = attack_ref.PhysicalTransRef(obj, mask, dist_range=attack_ref.TRAIN_DIST_RANGE)
    import random
    rnd = random.Random(seed)
    z0 = rnd.sample(pt.dist_range, n)
    al = rnd.sample(pt.angle_range, n)
    coeffs = [tv082.get_perspective_coeffs([list(map(float, p)) for p in pt.pos_obj_img_start],
                                           [list(map(float, p)) for p in pt.obj_pos_on_image(z0[i], al[i])])
              for i in range(n)]
    l_pad, t_pad = pt.pos_obj_img_start[0]
    return obj, mask, scenes, pt, z0, al, torch.tensor(coeffs, dtype=torch.float32), l_pad, t_pad


@pytest.mark.parametrize("n,seed,bcast,out_size", [(3, 1, False, (320, 1024)), (2, 2, True, (320, 1024)),
                                                   (2, 3, False, (330, 1100)),      # tiled kernel, ragged last tiles
                                                   (2, 4, False, (288, 960)),       # ratio 1.29: untiled 4-pixel kernel
                                                   (2, 5, False, (375, 1242))])     # OW % 4 != 0: generic kernel
def test_eot_paste_vs_oracle(n, seed, bcast, out_size):
    N, ops, _, attack_ref, synth, tv082 = _mods()
    obj, mask, scenes, pt, z0, al, coeffs, l_pad, t_pad = _paste_case(attack_ref, synth, tv082, n, seed)
    if bcast:
        scenes = scenes[:1]
    patch = obj.clone().requires_grad_(True)
    pt.reset_img(patch, mask)
    sc_full = scenes if not bcast else torch.cat(n * [scenes], 0)
    o_full, m_full, _, _ = pt.project(batch_size=n, z0_sample=z0, alpha_sample=al)      # phy_obj_atk.py:87-90
    adv, msk = tv082.resize(sc_full * (1 - m_full) + o_full * m_full, out_size), tv082.resize(m_full, out_size)
    gadv = torch.rand(adv.shape, generator=torch.Generator().manual_seed(9)) - 0.5
    (adv * gadv).sum().backward()
    dpatch = obj.cuda().requires_grad_(True)
    dadv, dmsk = ops.eot_paste(scenes.cuda(), dpatch, mask.cuda(), coeffs.cuda(), l_pad, t_pad, out_size)
    (dadv * gadv.cuda()).sum().backward()
    # the few elements over tolerance sit on the mask edge (a 0/1 step sampled through the homography); none is far off
    assert_close_frac(dadv, adv, rtol=1e-4, atol=2e-5, max_bad_frac=3e-5, name="adv scenes")
    assert_close_frac(dmsk, msk, rtol=1e-4, atol=2e-5, max_bad_frac=3e-5, name="mask out")
    assert float((dadv.detach().cpu() - adv.detach()).abs().max()) < 5e-4 and float((dmsk.cpu() - msk).abs().max()) < 5e-4
    assert float(dmsk.max()) > 0.99 and float(dmsk.min()) >= 0.0
    scale = patch.grad.abs().max().item()
    assert_close_frac(dpatch.grad, patch.grad, rtol=1e-3, atol=1e-4 * scale, max_bad_frac=1e-4, name="patch grad")


def test_eot_paste_properties():
    """Identity quad => the patch lands un-warped at its padded position; mask stays in [0,1]."""
    N, ops, _, attack_ref, synth, tv082 = _mods()
    obj, mask = synth.make_object()
    pt = attack_ref.PhysicalTransRef(obj, mask)
    start = [list(map(float, p)) for p in pt.pos_obj_img_start]
    coeffs = torch.tensor([tv082.get_perspective_coeffs(start, start)], dtype=torch.float32).cuda()
    scene = torch.zeros(1, 3, 375, 1242, device="cuda")
    l_pad, t_pad = pt.pos_obj_img_start[0]
    adv, msk = ops.eot_paste(scene, obj.cuda(), mask.cuda(), coeffs, l_pad, t_pad, (375, 1242))
    want = F.pad(obj * mask, [l_pad, 1242 - 300 - l_pad, t_pad, 375 - 260 - t_pad]).cuda()
    assert_close_frac(adv, want, rtol=1e-4, atol=2e-4, name="identity paste")
    assert float(msk.min()) >= 0 and float(msk.max()) <= 1 + 1e-6


def test_eot_paste_errors():
    N, ops, _, attack_ref, synth, tv082 = _mods()
    obj, mask = synth.make_object()
    with pytest.raises(RuntimeError, match="Batch size doesn't match"):
        ops.eot_paste(torch.zeros(3, 3, 375, 1242, device="cuda"), obj.cuda(), mask.cuda(),
                      torch.zeros(2, 8, device="cuda"), 471, 57, (320, 1024))
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.eot_paste(torch.zeros(2, 3, 375, 1242), obj, mask, torch.zeros(2, 8), 471, 57, (320, 1024))


@pytest.mark.parametrize("n", [5, 3 * 260 * 300, 2 * 3 * 320 * 1024 + 3])
def test_pgd_linf_step(n):
    N, ops, *_ = _mods()
    g = torch.Generator().manual_seed(n)
    x0 = torch.rand(n, generator=g)
    x = (x0 + (torch.rand(n, generator=g) - 0.5) * 0.2).clamp(0, 1)
    gr = torch.randn(n, generator=g)
    gr[::7] = 0.0
    eps, alpha = 0.1, 0.02
    want = torch.clamp(x0 + torch.clamp(x + alpha * gr.sign() - x0, min=-eps, max=eps), min=0, max=1)
    got = ops.pgd_linf_step(x.cuda(), x0.cuda(), gr.cuda(), alpha, eps)
    assert torch.equal(got.cpu(), want)
    # unaligned views take the scalar path
    got2 = ops.pgd_linf_step(x.cuda()[1:], x0.cuda()[1:], gr.cuda()[1:], alpha, eps)
    assert torch.equal(got2.cpu(), want[1:])


def test_l0_ops_vs_oracle():
    N, ops, _, attack_ref, synth, _ = _mods()
    obj, _ = synth.make_object()
    g = torch.Generator().manual_seed(3)
    pos = (torch.rand(obj.shape, generator=g) * 1.4 - 0.2)
    neg = (torch.rand(obj.shape, generator=g) * 1.4 - 0.2)
    pos[:, :, :40] *= 0.002
    neg[:, :, :40] *= 0.002
    pos_r, neg_r = pos.clone().requires_grad_(True), neg.clone().requires_grad_(True)
    p_pos = torch.clamp(pos_r, 0.0, 1.0)
    p_neg = -torch.clamp(neg_r, 0.0, 1.0)
    adv = torch.clamp(obj + (p_pos + p_neg), 0.0, 1.0)
    l0 = attack_ref.cal_l0(p_pos, p_neg, 1 / 255.0)
    cost = attack_ref.l0_mask_cost(pos_r, neg_r)
    gadv = torch.rand(obj.shape, generator=g) - 0.5
    ((adv * gadv).sum() + 0.06 * cost).backward()

    dp, dn = pos.cuda().requires_grad_(True), neg.cuda().requires_grad_(True)
    dadv, cnt = ops.l0_compose(obj.cuda(), dp, dn)
    dcost = ops.l0_mask_cost(dp, dn)
    mw = torch.tensor(0.06, device="cuda")
    ((dadv * gadv.cuda()).sum() + mw * dcost).backward()
    assert torch.equal(dadv.cpu(), adv.detach())
    assert int(cnt.item()) == int(l0)
    assert abs(dcost.item() - cost.item()) <= 1e-6 * abs(cost.item())
    assert_close_frac(dp.grad, pos_r.grad, rtol=1e-5, atol=1e-9, name="g_pos")
    assert_close_frac(dn.grad, neg_r.grad, rtol=1e-5, atol=1e-9, name="g_neg")
    # finalisation (phy_obj_atk_l0.py:143-150)
    pp, pn = p_pos.detach().clone(), p_neg.detach().clone()
    pp[pp < 1 / 255.0] = 0
    pn[pn > -1 / 255.0] = 0
    fin = torch.clamp(obj + (pp + pn), 0.0, 1.0)
    dfin, _ = ops.l0_compose(obj.cuda(), dp.detach(), dn.detach(), finalize=True)
    assert torch.equal(dfin.cpu(), fin)


@pytest.mark.parametrize("with_mask", [True, False])
def test_masked_sq_mean(with_mask):
    N, ops, *_ = _mods()
    g = torch.Generator().manual_seed(8)
    d = torch.rand(3, 1, 320, 1024, generator=g, requires_grad=True)
    m = torch.rand(3, 1, 320, 1024, generator=g) if with_mask else None
    want = torch.nn.MSELoss()(d * m if with_mask else d, torch.zeros_like(d))
    (-want).backward()
    dd = d.detach().cuda().requires_grad_(True)
    got = ops.masked_sq_mean(dd, m.cuda() if with_mask else None)
    (-got).backward()
    assert abs(got.item() - want.item()) <= 2e-6 * want.item()
    assert_close_frac(dd.grad, d.grad, rtol=1e-5, atol=1e-12, name="sq-mean grad")


def test_full_size_properties():
    """BASELINE config-2 shape (B=32, 320x1024): size-independent properties of the fused loss."""
    N, ops, _, _, synth, _ = _mods()
    B, H, W = 32, 320, 1024
    g = torch.Generator(device="cuda").manual_seed(5)
    left = F.avg_pool2d(torch.rand(B, 3, H + 4, W + 4, device="cuda", generator=g), 5, 1).contiguous()
    K, inv_K = synth.make_intrinsics(B, H, W)
    K, inv_K = K.cuda(), inv_K.cuda()
    T = torch.eye(4, device="cuda").repeat(B, 1, 1)
    T[:, 0, 3] = -0.1
    colors = [left if s == 0 else F.avg_pool2d(left, 2 ** s).contiguous() for s in range(4)]
    # (1) source == target and zero baseline: warp is the identity, both branches tie, loss = smoothness only
    T0 = torch.eye(4, device="cuda").repeat(B, 1, 1)
    disps = [torch.full((B, 1, H >> s, W >> s), 0.3, device="cuda") for s in range(4)]
    out = ops.photometric_smooth_loss(left, [left], [T0], K, inv_K, disps, colors, noise=None, want_to_opt=True)
    assert float(out.to_opt[0].abs().max()) < PIX_ATOL     # SSIM(x,x) = 0 up to fp32 conditioning noise
    assert abs(float(out.fin[N.FIN_LOSS])) < PIX_ATOL      # constant disparity: zero smoothness too
    # (2) linearity in the upstream gradient + determinism (no atomics on this path)
    disps = [(0.02 + 0.2 * torch.rand(B, 1, H >> s, W >> s, device="cuda", generator=g)).requires_grad_(True)
             for s in range(4)]
    right = torch.roll(left, 8, 3).contiguous()

    def run(scale):
        for d in disps:
            d.grad = None
        o = ops.photometric_smooth_loss(left, [right], [T], K, inv_K, disps, colors, noise=None)
        (o.fin[N.FIN_LOSS] * scale).backward()
        return o.fin.detach().clone(), [d.grad.clone() for d in disps]
    f1, g1 = run(1.0)
    f1b, g1b = run(1.0)
    f3, g3 = run(3.0)
    assert torch.equal(f1, f1b) and all(torch.equal(a, b) for a, b in zip(g1, g1b)), "not bitwise reproducible"
    for a, b in zip(g1, g3):
        assert torch.isfinite(a).all()
        torch.testing.assert_close(b, 3 * a, rtol=1e-5, atol=1e-5 * a.abs().max().item())
    # (3) count of selected pixels is consistent with the masks
    o = ops.photometric_smooth_loss(left, [right], [T], K, inv_K, [d.detach() for d in disps], colors, noise=None)
    for s in range(4):
        assert abs(float(o.fin[N.FIN_COUNT_S + s]) - float((o.sel[s] > 0).sum())) <= 2.0


@pytest.mark.parametrize("out_size", [(320, 1024), (375, 1242), (64, 190), (330, 1100), (300, 1000), (288, 960)])
def test_eot_paste_flip_is_the_mirrored_paste(out_size):
    """flip[n] != 0: the whole composite of sample n is written mirrored (mono_dataset.py:222-225 on an un-flipped frame);
    both the 4-pixel-per-thread kernel (OW % 4 == 0) and the generic one; gradients follow."""
    N, ops, _, attack_ref, synth, tv082 = _mods()
    obj, mask, scenes, pt, z0, al, coeffs, l_pad, t_pad = _paste_case(attack_ref, synth, tv082, 3, 5)
    flip = torch.tensor([1, 0, 1], dtype=torch.int32).cuda()
    p0 = obj.cuda().requires_grad_(True)
    a0, m0 = ops.eot_paste(scenes.cuda(), p0, mask.cuda(), coeffs.cuda(), l_pad, t_pad, out_size)
    p1 = obj.cuda().requires_grad_(True)
    a1, m1 = ops.eot_paste(scenes.cuda(), p1, mask.cuda(), coeffs.cuda(), l_pad, t_pad, out_size, flip)
    for n in range(3):
        want_a, want_m = (a0[n].flip(2), m0[n].flip(2)) if int(flip[n]) else (a0[n], m0[n])
        assert torch.equal(a1[n], want_a) and torch.equal(m1[n], want_m)
    gadv = torch.rand(a0.shape, generator=torch.Generator().manual_seed(1)).cuda() - 0.5
    (a0 * gadv).sum().backward()
    gflip = gadv.clone()
    gflip[0], gflip[2] = gadv[0].flip(2), gadv[2].flip(2)
    (a1 * gflip).sum().backward()
    assert torch.equal(p0.grad, p1.grad)


def test_avg_pyramid_is_bitwise_aten_avg_pool():
    """ops.avg_pyramid (one pass, three levels) against F.avg_pool2d(x, 2 / 4 / 8): identical bits (the sums are formed in
    ATen's order), at the training resolution and a ragged batch / channel count."""
    from depthmodelhardening_amd import ops
    for shape in ((32, 3, 320, 1024), (3, 5, 24, 40), (1, 1, 8, 8)):
        x = torch.rand(*shape, generator=torch.Generator().manual_seed(4)).cuda()
        got = ops.avg_pyramid(x)
        for s, g in zip((1, 2, 3), got):
            assert torch.equal(g, F.avg_pool2d(x, 2 ** s)), (shape, s)
    with pytest.raises(RuntimeError):
        ops.avg_pyramid(torch.rand(1, 3, 20, 40).cuda())      # 20 is not a multiple of 8


def test_synthesis_from_the_pool_by_index_equals_the_copy_path(tmp_path):
    """The GPU-side prep_adv_data reads its frames out of the dataset's pool by index (K3's scene_index: no index_select /
    side-pick copies): same bits as pasting into gathered copies of the frames, for mixed sides and flips."""
    from depthmodelhardening_amd.datasets import SyntheticKITTIDataset, make_object
    from depthmodelhardening_amd.my_utils import to_device_async
    dev = torch.device("cuda")
    ds = SyntheticKITTIDataset(64, 192, [0, "s"], 4, 8, dev, seed=3, pool=6)
    obj, mask = make_object(dev)
    # (the attack object of set_adv_train is not needed for the synthesis itself: set what synthesize() reads)
    from depthmodelhardening_amd.physicalTrans import PhysicalTrans
    from depthmodelhardening_amd.my_utils import ori_H, ori_W, train_dist_range
    ds.is_adv_train, ds.obj_mask, ds.obj_img_ben, ds.obj_img_adv = True, mask, obj, (obj * 0.9).contiguous()
    ds.ben_trans = PhysicalTrans(ds.obj_img_ben, mask, {'path': None}, (1, 3, ori_H, ori_W), dist_range=train_dist_range)
    ds.adv_trans = PhysicalTrans(ds.obj_img_adv, mask, {'path': None}, (1, 3, ori_H, ori_W), dist_range=train_dist_range)
    ds.adv_K = ds.K.copy()
    ds.adv_K[0, :] *= ori_W
    ds.adv_K[1, :] *= ori_H
    picks = [4, 0, 5, 2]
    geo = {"side": ["l", "r", "r", "l"], "flip": [False, True, False, True], "synth": [True, True, False, True],
           "z0": [5.0, 7.4, 9.8, 6.2], "alpha": [0, -15, 30, 10]}
    idx = to_device_async(picks, dev, torch.int64)
    ref = ds.synthesize(ds.raw_left.index_select(0, idx), ds.raw_right.index_select(0, idx), geo, (64, 192))
    P = ds.pool_size
    i0 = to_device_async([p + (0 if sd == "l" else P) for p, sd in zip(picks, geo["side"])], dev, torch.int32)
    i_s = to_device_async([p + (P if sd == "l" else 0) for p, sd in zip(picks, geo["side"])], dev, torch.int32)
    got = ds.synthesize(None, None, geo, (64, 192), pool_index=(i0, i_s))
    for a, b in zip(ref, got):
        assert torch.equal(a, b)

