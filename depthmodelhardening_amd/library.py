"""``torch.ops.dmh.*`` -- the hot-path kernels registered with ``torch.library`` on top of the unchanged C ABI.

SURVEY.md section 8b asks for "PyTorch-ROCm custom ops": operators the dispatcher knows, with a fake-tensor (meta)
implementation, so that they can be traced, exported and compiled (``torch.compile`` / ``torch.export`` see an opaque
``dmh::...`` call with known output shapes instead of a ctypes call they cannot look into) and an autograd formula that is
itself a registered op.  Each operator below is a thin schema around the same launches ``ops.py`` makes through
``libdmh_hip.so`` (include/dmh_hip.h); ``ops.py``'s ``torch.autograd.Function`` wrappers remain the path the Trainer and the
attacks use (they carry the per-attack caches and the fused multi-launch nodes), and tests/test_gpu_library.py holds the two
to the same bits.  There is no CPU kernel behind any of them: the CUDA implementations reject CPU tensors like ``ops.py`` does.

    dmh::eot_paste            physicalTrans.py:156-165 + phy_obj_atk.py:88-90   (K3)   + dmh::eot_paste_bwd
    dmh::masked_sq_mean       phy_obj_atk.py:94, pgd_depth.py:68-70             (K6)   + dmh::masked_sq_mean_bwd
    dmh::gt_depth_mse         MD2/trainer.py:551-557 (--supervised_adv --gt_depth) (K6b)  + dmh::gt_depth_mse_bwd
    dmh::pgd_linf_step        phy_obj_atk.py:98-101, pgd_depth.py:76-78         (K4)
    dmh::l0_compose           phy_obj_atk_l0.py:94-99,43-52                     (K5)   + dmh::l0_compose_bwd
    dmh::l0_mask_cost         phy_obj_atk_l0.py:130-132                         (K5)   + dmh::l0_mask_cost_bwd
    dmh::photo_smooth_loss    MD2/trainer.py:472-523,539-674 (DH/trainer.py:638-741 with variant 1)   (K1 + K2 + finalise)
                                                                                       + dmh::photo_smooth_loss_bwd
    dmh::ssim_map             MD2/layers.py:223-253 SSIM.forward                (K1's window sums, stand-alone)
    dmh::smooth_loss          MD2/layers.py:207-220 get_smooth_loss             (K2 on one scale)
"""
import ctypes as C
from typing import List, Optional, Tuple

import torch
from torch.library import custom_op

from . import _native as N
from . import ops


def _cu(t):
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------------------------------------------- K3
@custom_op("dmh::eot_paste", mutates_args=())
def eot_paste(scene: torch.Tensor, patch: torch.Tensor, pmask: torch.Tensor, coeffs: torch.Tensor, l_pad: int, t_pad: int,
              oh: int, ow: int, flip: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    scene, patch, pmask, coeffs = _cu(scene), _cu(patch), _cu(pmask), _cu(coeffs)
    a = ops._paste_args(scene, patch, pmask, coeffs, l_pad, t_pad, oh, ow, N.PASTE_COMPOSITE, flip)
    adv = torch.empty((a.N, 3, oh, ow), device=scene.device, dtype=torch.float32)
    mask_out = torch.empty((a.N, 1, oh, ow), device=scene.device, dtype=torch.float32)
    N.check(N.lib().dmh_eot_paste_fwd(C.byref(a), N.ptr(adv), N.ptr(mask_out), N.stream()))
    return adv, mask_out


@eot_paste.register_fake
def _(scene, patch, pmask, coeffs, l_pad, t_pad, oh, ow, flip=None):
    n = coeffs.shape[0]
    return scene.new_empty((n, 3, oh, ow)), scene.new_empty((n, 1, oh, ow))


@custom_op("dmh::eot_paste_bwd", mutates_args=())
def eot_paste_bwd(scene: torch.Tensor, patch: torch.Tensor, pmask: torch.Tensor, coeffs: torch.Tensor, l_pad: int, t_pad: int,
                  oh: int, ow: int, flip: Optional[torch.Tensor], g_adv: torch.Tensor) -> torch.Tensor:
    scene, patch, pmask, coeffs, g_adv = _cu(scene), _cu(patch), _cu(pmask), _cu(coeffs), _cu(g_adv)
    a = ops._paste_args(scene, patch, pmask, coeffs, l_pad, t_pad, oh, ow, N.PASTE_COMPOSITE, flip)
    g_patch = torch.empty_like(patch)
    N.check(N.lib().dmh_eot_paste_bwd(C.byref(a), N.ptr(g_adv), N.ptr(g_patch), N.stream()))
    return g_patch


@eot_paste_bwd.register_fake
def _(scene, patch, pmask, coeffs, l_pad, t_pad, oh, ow, flip, g_adv):
    return torch.empty_like(patch)


def _paste_setup(ctx, inputs, output):
    scene, patch, pmask, coeffs, l_pad, t_pad, oh, ow, flip = inputs
    ctx.save_for_backward(scene, patch, pmask, coeffs, *([flip] if flip is not None else []))
    ctx.geo = (l_pad, t_pad, oh, ow, flip is not None)


def _paste_backward(ctx, g_adv, g_mask):
    sv = ctx.saved_tensors
    l_pad, t_pad, oh, ow, has_flip = ctx.geo
    g_patch = torch.ops.dmh.eot_paste_bwd(sv[0], sv[1], sv[2], sv[3], l_pad, t_pad, oh, ow, sv[4] if has_flip else None, g_adv)
    return None, g_patch, None, None, None, None, None, None, None


eot_paste.register_autograd(_paste_backward, setup_context=_paste_setup)


# ----------------------------------------------------------------------------------------------------------------- K6
@custom_op("dmh::masked_sq_mean", mutates_args=())
def masked_sq_mean(disp: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    disp = _cu(disp)
    mask = None if mask is None else _cu(mask)
    lib = N.lib()
    n = disp.numel()
    if mask is not None and mask.numel() != n:
        raise RuntimeError("masked_sq_mean: mask and disp sizes differ")
    part = torch.empty(lib.dmh_sq_mean_partials_size(n), device=disp.device, dtype=torch.float32)
    cost = torch.empty((), device=disp.device, dtype=torch.float32)
    N.check(lib.dmh_masked_sq_mean_fwd(N.ptr(disp), N.ptr(mask), n, N.ptr(part), N.ptr(cost), N.stream()))
    return cost


@masked_sq_mean.register_fake
def _(disp, mask=None):
    return disp.new_empty(())


@custom_op("dmh::masked_sq_mean_bwd", mutates_args=())
def masked_sq_mean_bwd(disp: torch.Tensor, mask: Optional[torch.Tensor], g: torch.Tensor) -> torch.Tensor:
    disp = _cu(disp)
    mask = None if mask is None else _cu(mask)
    g_disp = torch.empty_like(disp)
    N.check(N.lib().dmh_masked_sq_mean_bwd(N.ptr(disp), N.ptr(mask), disp.numel(), N.ptr(_cu(g.to(torch.float32))), N.ptr(g_disp),
                                           N.stream()))
    return g_disp


@masked_sq_mean_bwd.register_fake
def _(disp, mask, g):
    return torch.empty_like(disp)


def _msm_setup(ctx, inputs, output):
    disp, mask = inputs
    ctx.save_for_backward(disp, *([mask] if mask is not None else []))
    ctx.has_mask = mask is not None


def _msm_backward(ctx, g):
    sv = ctx.saved_tensors
    return torch.ops.dmh.masked_sq_mean_bwd(sv[0], sv[1] if ctx.has_mask else None, g), None


masked_sq_mean.register_autograd(_msm_backward, setup_context=_msm_setup)


# ---------------------------------------------------------------------------------------------------------------- K6b
@custom_op("dmh::gt_depth_mse", mutates_args=())
def gt_depth_mse(disp: torch.Tensor, disp_gt: torch.Tensor, objmask: torch.Tensor, objdepth: torch.Tensor, min_depth: float,
                 max_depth: float) -> torch.Tensor:
    disp, disp_gt, objmask, bstride, objdepth, B, HW = ops._gt_depth_operands(disp, disp_gt, objmask, objdepth)
    lib = N.lib()
    part = torch.empty(lib.dmh_sq_mean_partials_size(B * HW), device=disp.device, dtype=torch.float32)
    cost = torch.empty((), device=disp.device, dtype=torch.float32)
    N.check(lib.dmh_gt_depth_mse_fwd(N.ptr(disp), N.ptr(disp_gt), C.c_void_p(objmask.data_ptr()), bstride, N.ptr(objdepth), B, HW,
                                     float(min_depth), float(max_depth), N.ptr(part), N.ptr(cost), N.stream()))
    return cost


@gt_depth_mse.register_fake
def _(disp, disp_gt, objmask, objdepth, min_depth, max_depth):
    return disp.new_empty(())


@custom_op("dmh::gt_depth_mse_bwd", mutates_args=())
def gt_depth_mse_bwd(disp: torch.Tensor, disp_gt: torch.Tensor, objmask: torch.Tensor, objdepth: torch.Tensor, min_depth: float,
                     max_depth: float, g: torch.Tensor) -> torch.Tensor:
    disp, disp_gt, objmask, bstride, objdepth, B, HW = ops._gt_depth_operands(disp, disp_gt, objmask, objdepth)
    g_disp = torch.empty_like(disp)
    N.check(N.lib().dmh_gt_depth_mse_bwd(N.ptr(disp), N.ptr(disp_gt), C.c_void_p(objmask.data_ptr()), bstride, N.ptr(objdepth), B, HW,
                                         float(min_depth), float(max_depth), N.ptr(_cu(g.to(torch.float32))), N.ptr(g_disp),
                                         N.stream()))
    return g_disp


@gt_depth_mse_bwd.register_fake
def _(disp, disp_gt, objmask, objdepth, min_depth, max_depth, g):
    return torch.empty_like(disp)


def _gtd_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1], inputs[2], inputs[3])
    ctx.depths = (inputs[4], inputs[5])


def _gtd_backward(ctx, g):
    disp, disp_gt, objmask, objdepth = ctx.saved_tensors
    return torch.ops.dmh.gt_depth_mse_bwd(disp, disp_gt, objmask, objdepth, ctx.depths[0], ctx.depths[1], g), None, None, None, None, None


gt_depth_mse.register_autograd(_gtd_backward, setup_context=_gtd_setup)


# ----------------------------------------------------------------------------------------------------------------- K4
@custom_op("dmh::pgd_linf_step", mutates_args=())
def pgd_linf_step(x: torch.Tensor, x0: torch.Tensor, grad: torch.Tensor, alpha: float, eps: float) -> torch.Tensor:
    x, x0, grad = _cu(x), _cu(x0), _cu(grad)
    if x.shape != x0.shape or x.shape != grad.shape:
        raise RuntimeError("pgd_linf_step: shape mismatch")
    out = torch.empty_like(x)
    N.check(N.lib().dmh_pgd_linf_step(N.ptr(x), N.ptr(x0), N.ptr(grad), float(alpha), float(eps), N.ptr(out), x.numel(),
                                      N.stream()))
    return out


@pgd_linf_step.register_fake
def _(x, x0, grad, alpha, eps):
    return torch.empty_like(x)


# ----------------------------------------------------------------------------------------------------------------- K5
@custom_op("dmh::l0_compose", mutates_args=())
def l0_compose(obj: torch.Tensor, pos: torch.Tensor, neg: torch.Tensor, l0_clip: float, finalize: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    obj, pos, neg = _cu(obj), _cu(pos), _cu(neg)
    if obj.dim() != 4 or obj.shape[0] != 1 or obj.shape != pos.shape or obj.shape != neg.shape:
        raise RuntimeError("l0_compose: obj/pos/neg must all be [1,C,H,W]")
    adv = torch.empty_like(obj)
    count = torch.zeros(1, device=obj.device, dtype=torch.int32)
    N.check(N.lib().dmh_l0_compose_fwd(N.ptr(obj), N.ptr(pos), N.ptr(neg), obj.shape[1], obj.shape[2] * obj.shape[3],
                                       float(l0_clip), int(finalize), N.ptr(adv), N.ptr(count), N.stream()))
    return adv, count


@l0_compose.register_fake
def _(obj, pos, neg, l0_clip, finalize):
    return torch.empty_like(obj), obj.new_empty((1,), dtype=torch.int32)


@custom_op("dmh::l0_compose_bwd", mutates_args=())
def l0_compose_bwd(obj: torch.Tensor, pos: torch.Tensor, neg: torch.Tensor, g_adv: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    obj, pos, neg, g_adv = _cu(obj), _cu(pos), _cu(neg), _cu(g_adv)
    g_pos, g_neg = torch.empty_like(pos), torch.empty_like(neg)
    N.check(N.lib().dmh_l0_compose_bwd(N.ptr(obj), N.ptr(pos), N.ptr(neg), N.ptr(g_adv), obj.shape[1], obj.shape[2] * obj.shape[3],
                                       N.ptr(g_pos), N.ptr(g_neg), 0, N.stream()))
    return g_pos, g_neg


@l0_compose_bwd.register_fake
def _(obj, pos, neg, g_adv):
    return torch.empty_like(pos), torch.empty_like(neg)


def _l0c_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1], inputs[2])


def _l0c_backward(ctx, g_adv, g_count):
    obj, pos, neg = ctx.saved_tensors
    g_pos, g_neg = torch.ops.dmh.l0_compose_bwd(obj, pos, neg, g_adv)
    return None, g_pos, g_neg, None, None


l0_compose.register_autograd(_l0c_backward, setup_context=_l0c_setup)


@custom_op("dmh::l0_mask_cost", mutates_args=())
def l0_mask_cost(pos: torch.Tensor, neg: torch.Tensor) -> torch.Tensor:
    pos, neg = _cu(pos), _cu(neg)
    if pos.dim() != 4 or pos.shape[0] != 1 or pos.shape != neg.shape:
        raise RuntimeError("l0_mask_cost: pos/neg must be [1,C,H,W]")
    lib = N.lib()
    Cc, HW = pos.shape[1], pos.shape[2] * pos.shape[3]
    part = torch.empty(lib.dmh_l0_mask_partials_size(HW), device=pos.device, dtype=torch.float32)
    cost = torch.empty((), device=pos.device, dtype=torch.float32)
    N.check(lib.dmh_l0_mask_cost_fwd(N.ptr(pos), N.ptr(neg), Cc, HW, N.ptr(part), N.ptr(cost), N.stream()))
    return cost


@l0_mask_cost.register_fake
def _(pos, neg):
    return pos.new_empty(())


@custom_op("dmh::l0_mask_cost_bwd", mutates_args=())
def l0_mask_cost_bwd(pos: torch.Tensor, neg: torch.Tensor, g: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    pos, neg = _cu(pos), _cu(neg)
    g_pos, g_neg = torch.empty_like(pos), torch.empty_like(neg)
    N.check(N.lib().dmh_l0_mask_cost_bwd(N.ptr(pos), N.ptr(neg), pos.shape[1], pos.shape[2] * pos.shape[3],
                                         N.ptr(_cu(g.to(torch.float32))), None, N.ptr(g_pos), N.ptr(g_neg), 0, N.stream()))
    return g_pos, g_neg


@l0_mask_cost_bwd.register_fake
def _(pos, neg, g):
    return torch.empty_like(pos), torch.empty_like(neg)


def _l0m_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])


def _l0m_backward(ctx, g):
    pos, neg = ctx.saved_tensors
    return torch.ops.dmh.l0_mask_cost_bwd(pos, neg, g)


l0_mask_cost.register_autograd(_l0m_backward, setup_context=_l0m_setup)


# ------------------------------------------------------------------------------------------------------- K1 + K2 + finalise
def _loss_cfg(sources, disps, min_depth, max_depth, variant, automask, no_ssim, smooth_wt, noise_mode, seed, offset):
    return dict(F=len(sources), NS=len(disps), min_depth=float(min_depth), max_depth=float(max_depth),
                variant="dh" if variant == N.VARIANT_DH else "md2", automask=bool(automask), no_ssim=bool(no_ssim),
                smooth_wt=float(smooth_wt), want_to_opt=False, hints=False, noise_mode=int(noise_mode), seed=int(seed),
                offset=int(offset))


@custom_op("dmh::photo_smooth_loss", mutates_args=())
def photo_smooth_loss(target: torch.Tensor, sources: List[torch.Tensor], Ts: List[torch.Tensor], K: torch.Tensor,
                      inv_K: torch.Tensor, disps: List[torch.Tensor], colors: List[torch.Tensor], min_depth: float,
                      max_depth: float, variant: int, automask: bool, no_ssim: bool, smooth_wt: float, noise_mode: int,
                      seed: int, offset: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(fin [FIN_SIZE], packed selection bytes [B,H,W], smoothness statistics): fin[FIN_LOSS] = losses["loss"],
    fin[FIN_LOSS_S + s] = losses["loss/s"] (include/dmh_hip.h).  noise_mode: 0 none, 2 in-kernel Philox (seed, offset)."""
    if noise_mode not in (N.NOISE_NONE, N.NOISE_PHILOX):
        raise RuntimeError("dmh::photo_smooth_loss: noise_mode must be 0 (none) or 2 (Philox)")
    cfg = _loss_cfg(sources, disps, min_depth, max_depth, variant, automask, no_ssim, smooth_wt, noise_mode, seed, offset)
    target, K, inv_K = _cu(target), _cu(K), _cu(inv_K)
    sources, Ts, disps, colors = [_cu(t) for t in sources], [_cu(t) for t in Ts], [_cu(t) for t in disps], [_cu(t) for t in colors]
    lib = N.lib()
    B, _, H, W = target.shape
    dev = target.device
    NS = len(disps)
    a = ops._photo_args(cfg, target, sources, Ts, K, inv_K, disps, ())
    sm = ops._smooth_args(disps, colors)
    sel = torch.empty((B, H, W), device=dev, dtype=torch.uint8)
    pp = torch.empty(lib.dmh_photo_partials_size(B, H, W, NS), device=dev, dtype=torch.float32)
    sp = torch.empty(lib.dmh_smooth_partials_size(C.byref(sm)), device=dev, dtype=torch.float32)
    fin = torch.empty(N.FIN_SIZE, device=dev, dtype=torch.float32)
    sstats = torch.empty((NS, B, 2), device=dev, dtype=torch.float32)
    st = N.stream()
    N.check(lib.dmh_photo_loss_fwd(C.byref(a), N.ptr(sel), N.ptr_array([None] * NS), N.ptr(pp), st))
    N.check(lib.dmh_smooth_loss_fwd(C.byref(sm), N.ptr(sp), st))
    N.check(lib.dmh_loss_finalize(N.ptr(pp), N.ptr(sp), B, H, W, C.byref(sm), a.variant, cfg["smooth_wt"], N.ptr(fin),
                                  N.ptr(sstats), st))
    return fin, sel, sstats


@photo_smooth_loss.register_fake
def _(target, sources, Ts, K, inv_K, disps, colors, min_depth, max_depth, variant, automask, no_ssim, smooth_wt, noise_mode,
      seed, offset):
    B, _, H, W = target.shape
    return (target.new_empty((N.FIN_SIZE,)), target.new_empty((B, H, W), dtype=torch.uint8),
            target.new_empty((len(disps), B, 2)))


@custom_op("dmh::photo_smooth_loss_bwd", mutates_args=())
def photo_smooth_loss_bwd(target: torch.Tensor, sources: List[torch.Tensor], Ts: List[torch.Tensor], K: torch.Tensor,
                          inv_K: torch.Tensor, disps: List[torch.Tensor], colors: List[torch.Tensor], min_depth: float,
                          max_depth: float, variant: int, automask: bool, no_ssim: bool, smooth_wt: float, fin: torch.Tensor,
                          sel: torch.Tensor, sstats: torch.Tensor, g_fin: torch.Tensor) -> List[torch.Tensor]:
    cfg = _loss_cfg(sources, disps, min_depth, max_depth, variant, automask, no_ssim, smooth_wt, N.NOISE_NONE, 0, 0)
    target, K, inv_K = _cu(target), _cu(K), _cu(inv_K)
    sources, Ts, disps, colors = [_cu(t) for t in sources], [_cu(t) for t in Ts], [_cu(t) for t in disps], [_cu(t) for t in colors]
    lib = N.lib()
    a = ops._photo_args(cfg, target, sources, Ts, K, inv_K, disps, ())
    sm = ops._smooth_args(disps, colors)
    gvec = _cu(g_fin.to(torch.float32))
    g_disp = [torch.empty_like(d) for d in disps]
    stage = torch.empty(lib.dmh_photo_stage_size(C.byref(a)), device=target.device, dtype=torch.float32)
    gp = N.ptr_array(g_disp)
    st = N.stream()
    N.check(lib.dmh_photo_loss_bwd(C.byref(a), N.ptr(_cu(sel)), N.ptr(gvec), N.ptr(_cu(fin)), N.ptr(stage), gp, st))
    N.check(lib.dmh_smooth_loss_bwd(C.byref(sm), N.ptr(gvec), N.ptr(_cu(sstats)), cfg["smooth_wt"], gp, 1, st))
    return g_disp


@photo_smooth_loss_bwd.register_fake
def _(target, sources, Ts, K, inv_K, disps, colors, min_depth, max_depth, variant, automask, no_ssim, smooth_wt, fin, sel, sstats,
      g_fin):
    return [torch.empty_like(d) for d in disps]


def _psl_setup(ctx, inputs, output):
    (target, sources, Ts, K, inv_K, disps, colors, min_depth, max_depth, variant, automask, no_ssim, smooth_wt, _nm, _seed,
     _off) = inputs
    fin, sel, sstats = output
    ctx.nf, ctx.ns = len(sources), len(disps)
    ctx.save_for_backward(target, K, inv_K, fin, sel, sstats, *sources, *Ts, *disps, *colors)
    ctx.scalars = (min_depth, max_depth, variant, automask, no_ssim, smooth_wt)


def _psl_backward(ctx, g_fin, g_sel, g_sstats):
    sv = ctx.saved_tensors
    target, K, inv_K, fin, sel, sstats = sv[:6]
    nf, ns = ctx.nf, ctx.ns
    sources, Ts = list(sv[6:6 + nf]), list(sv[6 + nf:6 + 2 * nf])
    disps, colors = list(sv[6 + 2 * nf:6 + 2 * nf + ns]), list(sv[6 + 2 * nf + ns:6 + 2 * nf + 2 * ns])
    g_disp = torch.ops.dmh.photo_smooth_loss_bwd(target, sources, Ts, K, inv_K, disps, colors, *ctx.scalars, fin, sel, sstats,
                                                 g_fin)
    # one entry per input, list inputs as lists of the same length
    return (None, [None] * nf, [None] * nf, None, None, list(g_disp), [None] * ns) + (None,) * 9


photo_smooth_loss.register_autograd(_psl_backward, setup_context=_psl_setup)


# ------------------------------------------------------------------------- the stand-alone layers surface (MD2/layers.py)
@custom_op("dmh::ssim_map", mutates_args=())
def ssim_map(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """clamp((1 - SSIM(x, y)) / 2, 0, 1) per channel with 3x3 reflection-padded means: MD2/layers.py:223-253."""
    x, y = _cu(x), _cu(y)
    if x.dim() != 4 or x.shape != y.shape:
        raise RuntimeError("ssim_map: x and y must be [B,C,H,W] of the same shape")
    B, Cc, H, W = x.shape
    out = torch.empty_like(x)
    N.check(N.lib().dmh_ssim_map(N.ptr(x), N.ptr(y), B * Cc, H, W, N.ptr(out), N.stream()))
    return out


@ssim_map.register_fake
def _(x, y):
    return torch.empty_like(x)


@custom_op("dmh::ssim_map_bwd", mutates_args=())
def ssim_map_bwd(x: torch.Tensor, y: torch.Tensor, g_out: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    x, y, g_out = _cu(x), _cu(y), _cu(g_out)
    B, Cc, H, W = x.shape
    ws = torch.empty(5 * x.numel(), device=x.device, dtype=torch.float32)
    g_x, g_y = torch.empty_like(x), torch.empty_like(y)
    N.check(N.lib().dmh_ssim_map_bwd(N.ptr(x), N.ptr(y), N.ptr(g_out), B * Cc, H, W, N.ptr(ws), N.ptr(g_x), N.ptr(g_y),
                                     N.stream()))
    return g_x, g_y


@ssim_map_bwd.register_fake
def _(x, y, g_out):
    return torch.empty_like(x), torch.empty_like(y)


def _ssim_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])


def _ssim_backward(ctx, g):
    x, y = ctx.saved_tensors
    g_x, g_y = torch.ops.dmh.ssim_map_bwd(x, y, g)
    return g_x, g_y


ssim_map.register_autograd(_ssim_backward, setup_context=_ssim_setup)


@custom_op("dmh::smooth_loss", mutates_args=())
def smooth_loss(disp: torch.Tensor, img: torch.Tensor) -> torch.Tensor:
    """mean(|d_x disp| exp(-mean_c |d_x img|)) + mean(|d_y disp| exp(-mean_c |d_y img|)): MD2/layers.py:207-220 on the
    disparity as given (the caller applies the mean normalisation of MD2/trainer.py:662-664)."""
    disp, img = _cu(disp), _cu(img)
    if disp.dim() != 4 or disp.shape[1] != 1 or img.dim() != 4 or img.shape[0] != disp.shape[0] or img.shape[2:] != disp.shape[2:]:
        raise RuntimeError("smooth_loss: disp [B,1,H,W] and img [B,C,H,W] expected")
    B, Cc, H, W = img.shape
    lib = N.lib()
    part = torch.empty(lib.dmh_edge_smooth_partials_size(B, H, W), device=disp.device, dtype=torch.float32)
    out = torch.empty((), device=disp.device, dtype=torch.float32)
    N.check(lib.dmh_edge_smooth(N.ptr(disp), N.ptr(img), B, Cc, H, W, N.ptr(part), N.ptr(out), N.stream()))
    return out


@smooth_loss.register_fake
def _(disp, img):
    return disp.new_empty(())


@custom_op("dmh::smooth_loss_bwd", mutates_args=())
def smooth_loss_bwd(disp: torch.Tensor, img: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    disp, img = _cu(disp), _cu(img)
    B, Cc, H, W = img.shape
    g_disp = torch.empty_like(disp)
    N.check(N.lib().dmh_edge_smooth_bwd(N.ptr(disp), N.ptr(img), B, Cc, H, W, N.ptr(_cu(g.to(torch.float32))), N.ptr(g_disp),
                                        N.stream()))
    return g_disp


@smooth_loss_bwd.register_fake
def _(disp, img, g):
    return torch.empty_like(disp)


def _smooth_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])


def _smooth_backward(ctx, g):
    disp, img = ctx.saved_tensors
    return torch.ops.dmh.smooth_loss_bwd(disp, img, g), None      # the image is data: no gradient flows to it here


smooth_loss.register_autograd(_smooth_backward, setup_context=_smooth_setup)


OPS = ("eot_paste", "eot_paste_bwd", "masked_sq_mean", "masked_sq_mean_bwd", "gt_depth_mse", "gt_depth_mse_bwd", "pgd_linf_step", "l0_compose", "l0_compose_bwd",
       "l0_mask_cost", "l0_mask_cost_bwd", "photo_smooth_loss", "photo_smooth_loss_bwd", "ssim_map", "ssim_map_bwd", "smooth_loss", "smooth_loss_bwd")
