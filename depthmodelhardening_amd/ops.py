"""Autograd-aware Python faces of the HIP kernels in libdmh_hip.so.

PyTorch is plumbing here: it owns device memory, streams and the autograd graph; every op
below is one or a few launches of hand-written gfx950 kernels through the C ABI
(include/dmh_hip.h).  There is no eager/CPU fallback: CPU tensors raise RuntimeError.
"""
import contextlib
import ctypes as C
import os
from collections import namedtuple

import torch

from . import _native as N

LossOut = namedtuple("LossOut", ["fin", "sel", "to_opt"])

# Optional per-launch HIP-event timing of the K1 kernels (bench.py's roofline figure): a dict
# name -> list of (start, end) torch.cuda.Event pairs recorded on the launch stream, or None (off).
_profile = None
_profile_only = None


def enable_profile(on=True, only=None):
    """HIP-event timing of the instrumented launches.  ``only``: tuple of name prefixes to time (the others launch
    untouched) -- an event pair adds ~10 us of dispatch latency around a launch, so a timed region should carry events
    on the kernels it reports and nothing else."""
    global _profile, _profile_only
    _profile = {} if on else None
    _profile_only = tuple(only) if (on and only) else None


def profile_ms():
    """Average launch duration in ms per instrumented kernel (synchronises)."""
    if not _profile:
        return {}
    torch.cuda.synchronize()
    return {k: sum(e[0].elapsed_time(e[1]) for e in v) / len(v) for k, v in _profile.items() if v}


def profile_bytes():
    """Per instrumented kernel: (launches, total ms, total algorithmic bytes, total direct-convolution flops) -- for
    kernels whose shape varies from launch to launch."""
    if not _profile:
        return {}
    torch.cuda.synchronize()
    return {k: (len(v), sum(e[0].elapsed_time(e[1]) for e in v), sum(e[2] for e in v), sum(e[3] for e in v))
            for k, v in _profile.items() if v}


def profiling_every_launch():
    """True while enable_profile(True) without ``only`` is in force: every instrumented launch carries an event pair (bench.py's
    one fully instrumented step).  Events cannot be read back from a captured HIP graph, so the attack runs that step eagerly."""
    return _profile is not None and _profile_only is None


def _timed(name, launch, nbytes=0, flops=0):
    if _profile is None or (_profile_only is not None and not name.startswith(_profile_only)):
        return launch()
    if torch.cuda.is_current_stream_capturing():      # an event recorded into a graph has no time stamp to read
        return launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = launch()
    e1.record()
    _profile.setdefault(name, []).append((e0, e1, nbytes, flops))
    return rc


def _philox(device, count):
    """(seed, offset) for the in-kernel tie-break noise, drawn from torch's CUDA generator state so
    that torch.manual_seed() makes runs reproducible; the generator offset is advanced past the
    ``count`` counters this call consumes."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    gen = torch.cuda.default_generators[idx]
    seed, off = gen.initial_seed(), gen.get_offset()
    gen.set_offset(off + ((count + 3) // 4) * 4)
    return seed & 0xFFFFFFFFFFFFFFFF, off


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _photo_args(cfg, target, sources, Ts, K, inv_K, disps, noises, hint=None):
    a = N.PhotoArgs()
    B, _, H, W = target.shape
    if hint is not None:
        for t_ in hint:
            if tuple(t_.shape) != (B, 1, H, W):
                raise RuntimeError("photometric loss: depth_hint / depth_hint_mask must be [B,1,H,W]")
        a.depth_hint, a.depth_hint_mask = N.ptr(hint[0]), N.ptr(hint[1])
    a.target = N.ptr(target)
    for f, (s_, t_) in enumerate(zip(sources, Ts)):
        if tuple(s_.shape) != (B, 3, H, W) or tuple(t_.shape) != (B, 4, 4):
            raise RuntimeError("photometric loss: source/T shape mismatch %s %s" % (tuple(s_.shape), tuple(t_.shape)))
        a.source[f] = N.ptr(s_)
        a.T[f] = N.ptr(t_)
    if tuple(K.shape) != (B, 4, 4) or tuple(inv_K.shape) != (B, 4, 4):
        raise RuntimeError("photometric loss: K / inv_K must be [B,4,4]")
    a.K, a.inv_K = N.ptr(K), N.ptr(inv_K)
    for s, d in enumerate(disps):
        if d.dim() != 4 or d.shape[0] != B or d.shape[1] != 1:
            raise RuntimeError("photometric loss: disp[%d] must be [B,1,Hs,Ws]" % s)
        a.disp[s] = N.ptr(d)
        a.Hs[s], a.Ws[s] = d.shape[2], d.shape[3]
    a.B, a.H, a.W = B, H, W
    a.num_frames, a.num_scales = len(sources), len(disps)
    a.min_depth, a.max_depth = cfg["min_depth"], cfg["max_depth"]
    a.variant = N.VARIANT_MD2 if cfg["variant"] == "md2" else N.VARIANT_DH
    a.automask, a.no_ssim = int(cfg["automask"]), int(cfg["no_ssim"])
    a.noise_mode = cfg["noise_mode"]
    if cfg["noise_mode"] == N.NOISE_TENSOR:
        nf = len(sources) if cfg["variant"] == "md2" else 1
        for s, z in enumerate(noises):
            if tuple(z.shape) != (B, nf, H, W):
                raise RuntimeError("photometric loss: noise[%d] must be [B,%d,H,W]" % (s, nf))
            a.noise[s] = N.ptr(z)
    a.seed, a.offset = cfg.get("seed", 0), cfg.get("offset", 0)
    return a


def _smooth_args(disps, colors):
    a = N.SmoothArgs()
    B = disps[0].shape[0]
    for s, (d, c) in enumerate(zip(disps, colors)):
        if tuple(c.shape) != (B, 3, d.shape[2], d.shape[3]):
            raise RuntimeError("smoothness: color[%d] %s does not match disp %s" % (s, tuple(c.shape), tuple(d.shape)))
        a.disp[s], a.color[s] = N.ptr(d), N.ptr(c)
        a.Hs[s], a.Ws[s] = d.shape[2], d.shape[3]
    a.B, a.num_scales = B, len(disps)
    return a


class _PhotoSmoothLoss(torch.autograd.Function):
    """K1 + K2 + finalise.  Tensor args: target, K, inv_K, then F sources, F Ts, NS colors,
    (NS noises), NS disps."""

    @staticmethod
    def forward(ctx, cfg, target, K, inv_K, *rest):
        F, NS = cfg["F"], cfg["NS"]
        sources, Ts = rest[:F], rest[F:2 * F]
        colors = rest[2 * F:2 * F + NS]
        pos = 2 * F + NS
        noises = ()
        if cfg["noise_mode"] == N.NOISE_TENSOR:
            noises = rest[pos:pos + NS]
            pos += NS
        disps = rest[pos:pos + NS]
        hint = tuple(rest[pos + NS:pos + NS + 2]) if cfg["hints"] else None
        lib = N.lib()
        B, _, H, W = target.shape
        dev = target.device
        a = _photo_args(cfg, target, sources, Ts, K, inv_K, disps, noises, hint)
        sm = _smooth_args(disps, colors)
        sel = torch.empty((B, H, W), device=dev, dtype=torch.uint8)    # 2 bits per scale: 0 identity, 1+f frame f
        to_opt = [torch.empty((B, H, W), device=dev, dtype=torch.float32) if cfg["want_to_opt"] else None
                  for _ in range(NS)]
        pp = torch.empty(lib.dmh_photo_partials_size(B, H, W, NS), device=dev, dtype=torch.float32)
        sp = torch.empty(lib.dmh_smooth_partials_size(C.byref(sm)), device=dev, dtype=torch.float32)
        fin = torch.empty(N.FIN_SIZE, device=dev, dtype=torch.float32)
        sstats = torch.empty((NS, B, 2), device=dev, dtype=torch.float32)
        st = N.stream()
        optp = N.ptr_array(to_opt)
        N.check(_timed("photo_fwd", lambda: lib.dmh_photo_loss_fwd(C.byref(a), N.ptr(sel), optp, N.ptr(pp), st)))
        N.check(lib.dmh_smooth_loss_fwd(C.byref(sm), N.ptr(sp), st))
        N.check(lib.dmh_loss_finalize(N.ptr(pp), N.ptr(sp), B, H, W, C.byref(sm), a.variant, cfg["smooth_wt"],
                                      N.ptr(fin), N.ptr(sstats), st))
        ctx.cfg = cfg
        ctx.save_for_backward(target, K, inv_K, fin, sstats, sel, *sources, *Ts, *colors, *disps, *(hint or ()))
        outs = [fin, sel] + [t for t in to_opt if t is not None]
        ctx.mark_non_differentiable(*outs[1:])
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_fin, *unused):
        cfg = ctx.cfg
        F, NS = cfg["F"], cfg["NS"]
        sv = ctx.saved_tensors
        target, K, inv_K, fin, sstats, sel = sv[:6]
        sources, Ts = sv[6:6 + F], sv[6 + F:6 + 2 * F]
        colors = sv[6 + 2 * F:6 + 2 * F + NS]
        disps = sv[6 + 2 * F + NS:6 + 2 * F + 2 * NS]
        hint = tuple(sv[6 + 2 * F + 2 * NS:6 + 2 * F + 2 * NS + 2]) if cfg["hints"] else None
        lib = N.lib()
        dev = target.device
        gvec = _c(g_fin.to(torch.float32))
        cfg_b = dict(cfg, noise_mode=N.NOISE_NONE)
        a = _photo_args(cfg_b, target, sources, Ts, K, inv_K, disps, (), hint)
        sm = _smooth_args(disps, colors)
        st = N.stream()
        g_disp = [torch.empty_like(d) for d in disps]
        stage = torch.empty(lib.dmh_photo_stage_size(C.byref(a)), device=dev, dtype=torch.float32)
        gp = N.ptr_array(g_disp)
        want_pose = [bool(ctx.needs_input_grad[4 + F + f]) for f in range(F)]      # cam_T_cam of source frame f
        g_T = [None] * F
        if any(want_pose):
            # monocular frames: K1's backward also leaves twelve sums per (scale, strip, frame), from which
            # d loss / d P_f (P_f = (K T_f)[:3,:]) and d loss / d T_f = K[:3,:]^T dP_f follow (include/dmh_hip.h)
            B = target.shape[0]
            part = torch.empty(lib.dmh_photo_pose_partials_size(C.byref(a)), device=dev, dtype=torch.float32)
            N.check(_timed("photo_bwd", lambda: lib.dmh_photo_loss_bwd_pose(C.byref(a), N.ptr(sel), N.ptr(gvec), N.ptr(fin),
                                                                           N.ptr(stage), gp, N.ptr(part), st)))
            sums = part.view(NS, B, -1, F, 3, 4).double().sum((0, 2))               # [B, F, 3, (S_i0, S_i1, S_i2, s_i)]
            Kd, iKd = K.double(), inv_K.double()
            for f in range(F):
                if not want_pose[f]:
                    continue
                dP = torch.cat([torch.matmul(sums[:, f, :, :3], iKd[:, :3, :3].transpose(1, 2)), sums[:, f, :, 3:4]], 2)
                g_T[f] = torch.matmul(Kd[:, :3, :].transpose(1, 2), dP).to(Ts[f].dtype)     # [B, 4, 4]
        else:
            N.check(_timed("photo_bwd", lambda: lib.dmh_photo_loss_bwd(C.byref(a), N.ptr(sel), N.ptr(gvec), N.ptr(fin),
                                                                      N.ptr(stage), gp, st)))
        N.check(lib.dmh_smooth_loss_bwd(C.byref(sm), N.ptr(gvec), N.ptr(sstats), cfg["smooth_wt"], gp, 1, st))
        n_tail = NS + (NS if cfg["noise_mode"] == N.NOISE_TENSOR else 0)
        return ((None, None, None, None) + (None,) * F + tuple(g_T) + (None,) * n_tail + tuple(g_disp)
                + ((None, None) if cfg["hints"] else ()))


class SelectionMaps(object):
    """The per-scale selection maps of the fused loss, unpacked on demand from the packed byte map the kernel writes
    (2 bits per scale): ``maps[s]`` is a float [B,H,W] tensor, 0 where the identity term was chosen, 1 + f where the
    reprojection of source frame f was (== outputs["identity_selection/s"] for one source frame,
    MD2/trainer.py:656-658).  Nothing is materialised until a scale is asked for."""

    def __init__(self, packed, num_scales):
        self.packed, self.num_scales, self._cache = packed, num_scales, {}

    def __len__(self):
        return self.num_scales

    def __getitem__(self, s):
        if not 0 <= s < self.num_scales:
            raise IndexError(s)
        if s not in self._cache:
            out = torch.empty(self.packed.shape, device=self.packed.device, dtype=torch.float32)
            N.check(N.lib().dmh_unpack_selection(N.ptr(self.packed), self.packed.numel(), int(s), N.ptr(out), N.stream()))
            self._cache[s] = out
        return self._cache[s]

    def __iter__(self):
        return (self[s] for s in range(self.num_scales))


def photometric_smooth_loss(target, sources, Ts, K, inv_K, disps, colors, min_depth=0.1, max_depth=100.0,
                            variant="md2", automask=True, no_ssim=False, smooth_wt=1e-3, noise="philox",
                            want_to_opt=False, depth_hint=None, depth_hint_mask=None):
    """Fused photometric-reprojection + SSIM + auto-mask + smoothness loss over all scales.

    target [B,3,H,W]; sources / Ts: one per non-target frame; disps[s] [B,1,H/2^s,W/2^s];
    colors[s] = inputs[("color",0,s)].  ``noise``: "philox" (in-kernel randn*1e-5, the reference's
    tie-break of MD2/trainer.py:642-645), None, or a list of NS already-scaled tensors.
    ``depth_hint`` / ``depth_hint_mask`` [B,1,H,W]: DepthHints' --use_depth_hints (DH/trainer.py:510-525,629-636,
    700-725; variant "dh", one source frame): fin[FIN_HINT_S+s] is depth_hint_loss/s (already part of loss/s) and
    sel[s] == 3 marks the pixels where the hint won the argmin (outputs["depth_hint_pixels/s"]).
    Returns LossOut(fin, sel, to_opt): fin[FIN_*] is differentiable w.r.t. the disparities; sel is a SelectionMaps
    (sel[s] unpacks scale s on demand).
    """
    F, NS = len(sources), len(disps)
    N.ptr(target)   # rejects CPU tensors up front ("no CPU path") before any CUDA-only call below
    if not (1 <= F <= 3 and 1 <= NS <= N.MAX_SCALES):
        raise RuntimeError("photometric loss supports 1..3 source frames and 1..4 scales")
    if variant not in ("md2", "dh"):
        raise RuntimeError("variant must be 'md2' or 'dh'")
    B, _, H, W = target.shape
    if (depth_hint is None) != (depth_hint_mask is None):
        raise RuntimeError("depth_hint and depth_hint_mask go together")
    cfg = dict(F=F, NS=NS, min_depth=float(min_depth), max_depth=float(max_depth), variant=variant,
               automask=bool(automask), no_ssim=bool(no_ssim), smooth_wt=float(smooth_wt), want_to_opt=want_to_opt,
               hints=depth_hint is not None)
    noises = ()
    if noise is None or not automask:
        cfg["noise_mode"] = N.NOISE_NONE
    elif isinstance(noise, str):
        cfg["noise_mode"] = N.NOISE_PHILOX
        cfg["seed"], cfg["offset"] = _philox(target.device, NS * B * F * H * W)
    else:
        cfg["noise_mode"] = N.NOISE_TENSOR
        noises = tuple(_c(z) for z in noise)
    args = (_c(target), _c(K), _c(inv_K)) + tuple(_c(s) for s in sources) + tuple(_c(t) for t in Ts) + \
        tuple(_c(c) for c in colors) + noises + tuple(_c(d) for d in disps)
    if depth_hint is not None:
        args += (_c(depth_hint), _c(depth_hint_mask))
    outs = _PhotoSmoothLoss.apply(cfg, *args)
    fin, sel = outs[0], SelectionMaps(outs[1], NS)
    to_opt = list(outs[2:]) if want_to_opt else [None] * NS
    return LossOut(fin, sel, to_opt)


class _WarpView(torch.autograd.Function):
    @staticmethod
    def forward(ctx, source, disp, K, inv_K, T, H, W, min_depth, max_depth):
        lib = N.lib()
        B = source.shape[0]
        Hs, Ws = disp.shape[2], disp.shape[3]
        dev = source.device
        depth = torch.empty((B, 1, H, W), device=dev, dtype=torch.float32)
        sample = torch.empty((B, H, W, 2), device=dev, dtype=torch.float32)
        color = torch.empty((B, 3, H, W), device=dev, dtype=torch.float32)
        N.check(lib.dmh_warp_view_fwd(N.ptr(source), N.ptr(disp), N.ptr(K), N.ptr(inv_K), N.ptr(T), B, H, W, Hs, Ws,
                                      min_depth, max_depth, N.ptr(depth), N.ptr(sample), N.ptr(color), N.stream()))
        ctx.save_for_backward(source, disp, K, inv_K, T)
        ctx.geo = (H, W, min_depth, max_depth)
        ctx.mark_non_differentiable(sample)
        return depth, sample, color

    @staticmethod
    def backward(ctx, g_depth, g_sample, g_color):
        source, disp, K, inv_K, T = ctx.saved_tensors
        H, W, min_depth, max_depth = ctx.geo
        lib = N.lib()
        B = source.shape[0]
        Hs, Ws = disp.shape[2], disp.shape[3]
        g_up = torch.empty((B, H, W), device=source.device, dtype=torch.float32)
        gc = _c(g_color) if g_color is not None else None
        gd = _c(g_depth) if g_depth is not None else None
        st = N.stream()
        N.check(lib.dmh_warp_view_bwd(N.ptr(source), N.ptr(disp), N.ptr(K), N.ptr(inv_K), N.ptr(T), B, H, W, Hs, Ws,
                                      min_depth, max_depth, N.ptr(gc), N.ptr(gd), N.ptr(g_up), st))
        if (Hs, Ws) == (H, W):
            g = g_up.view(B, 1, H, W)
        else:
            g = torch.empty_like(disp)
            N.check(lib.dmh_upsample_bilinear_adjoint(N.ptr(g_up), N.ptr(g), B, H, W, Hs, Ws, 0, st))
        return None, g, None, None, None, None, None, None, None


def warp_view(source, disp, K, inv_K, T, H, W, min_depth=0.1, max_depth=100.0):
    """One (scale, frame) body of generate_images_pred (MD2/trainer.py:481-519):
    returns (depth [B,1,H,W], sample grid [B,H,W,2], warped colour [B,3,H,W])."""
    return _WarpView.apply(_c(source), _c(disp), _c(K), _c(inv_K), _c(T), int(H), int(W), float(min_depth),
                           float(max_depth))


def _paste_args(scene, patch, pmask, coeffs, l_pad, t_pad, OH, OW, mode=N.PASTE_COMPOSITE, flip=None, scene_index=None):
    a = N.PasteArgs()
    if flip is not None:
        if flip.dtype != torch.int32 or flip.numel() != coeffs.shape[0]:
            raise RuntimeError("eot_paste: flip must be an int32 tensor with one entry per sample")
        a.flip = N.ptr(flip)
    n = coeffs.shape[0]
    a.mode = mode
    if scene_index is not None:     # `scene` is a pool of frames; sample i reads frame scene_index[i] (range-checked by the caller)
        if scene_index.dtype != torch.int32 or scene_index.numel() != n or scene.shape[1] != 3:
            raise RuntimeError("eot_paste: scene_index must be an int32 tensor with one frame index per sample")
        a.scene_index = N.ptr(scene_index)
    elif scene.shape[0] not in (1, n) or scene.shape[1] != 3:
        raise RuntimeError("Batch size doesn't match!")
    if patch.dim() != 4 or patch.shape[0] != 1 or patch.shape[1] != 3 or tuple(pmask.shape) != (1, 1) + tuple(patch.shape[2:]):
        raise RuntimeError("eot_paste: patch must be [1,3,PH,PW] and mask [1,1,PH,PW]")
    if tuple(coeffs.shape) != (n, 8):
        raise RuntimeError("eot_paste: coeffs must be [N,8]")
    a.scene = N.ptr(scene) if mode == N.PASTE_COMPOSITE else None
    a.patch, a.pmask, a.coeffs = N.ptr(patch), N.ptr(pmask), N.ptr(coeffs)
    a.scene_bstride = 0 if (scene.shape[0] == 1 and scene_index is None) else scene.shape[1] * scene.shape[2] * scene.shape[3]
    a.N, a.SH, a.SW = n, scene.shape[2], scene.shape[3]
    a.PH, a.PW, a.OH, a.OW = patch.shape[2], patch.shape[3], OH, OW
    a.l_pad, a.t_pad = l_pad, t_pad
    return a


class _EotPaste(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scene, patch, pmask, coeffs, l_pad, t_pad, OH, OW, mode, flip=None, scene_index=None):
        lib = N.lib()
        a = _paste_args(scene, patch, pmask, coeffs, l_pad, t_pad, OH, OW, mode, flip, scene_index)
        adv = torch.empty((a.N, 3, OH, OW), device=scene.device, dtype=torch.float32)
        mask_out = torch.empty((a.N, 1, OH, OW), device=scene.device, dtype=torch.float32)
        # algorithmic bytes: the scene read once (one copy when it is broadcast) + patch and mask + the two outputs
        n_scene = scene.numel() if scene_index is None else a.N * 3 * a.SH * a.SW      # frames read, not the pool's size
        nb = 4 * (n_scene + patch.numel() + pmask.numel() + adv.numel() + mask_out.numel()) if mode == N.PASTE_COMPOSITE \
            else 4 * (patch.numel() + pmask.numel() + adv.numel() + mask_out.numel())
        N.check(_timed("paste_fwd", lambda: lib.dmh_eot_paste_fwd(C.byref(a), N.ptr(adv), N.ptr(mask_out), N.stream()), nb))
        ctx.save_for_backward(scene, patch, pmask, coeffs)
        ctx.geo = (l_pad, t_pad, OH, OW, mode)
        ctx.flip = flip
        ctx.scene_index = scene_index
        ctx.mark_non_differentiable(mask_out)
        return adv, mask_out

    @staticmethod
    def backward(ctx, g_adv, g_mask):
        scene, patch, pmask, coeffs = ctx.saved_tensors
        l_pad, t_pad, OH, OW, mode = ctx.geo
        lib = N.lib()
        a = _paste_args(scene, patch, pmask, coeffs, l_pad, t_pad, OH, OW, mode, ctx.flip, ctx.scene_index)
        g_patch = torch.empty_like(patch)
        g_adv = _c(g_adv)
        N.check(_timed("paste_bwd", lambda: lib.dmh_eot_paste_bwd(C.byref(a), N.ptr(g_adv), N.ptr(g_patch), N.stream()),
                       4 * (g_adv.numel() + g_patch.numel())))
        return None, g_patch, None, None, None, None, None, None, None, None, None


def eot_paste(scene, patch, pmask, coeffs, l_pad, t_pad, out_size, flip=None, scene_index=None):
    """Pad -> perspective(patch, mask) -> composite -> Resize, fused (physicalTrans.py:107-166 +
    phy_obj_atk.py:87-90).  Returns (adv [N,3,OH,OW], mask_out [N,1,OH,OW]); differentiable w.r.t. patch.
    ``flip`` (int32 [N], optional): samples to mirror horizontally (mono_dataset.py:222-225 on an un-flipped scene).
    ``scene_index`` (int32 [N], optional): ``scene`` is a POOL of frames [P,3,SH,SW] and sample i reads frame scene_index[i]
    -- the loader's choice of frames and camera sides without copying them (the caller guarantees 0 <= index < P)."""
    return _EotPaste.apply(_c(scene), _c(patch), _c(pmask), _c(coeffs), int(l_pad), int(t_pad), int(out_size[0]),
                           int(out_size[1]), N.PASTE_COMPOSITE, flip, scene_index)


def perspective_warp(patch, pmask, coeffs, l_pad, t_pad, frame_size):
    """Pad + torchvision-0.8.2 perspective of patch and mask into [N,3,SH,SW] / [N,1,SH,SW] frames
    (physicalTrans.py:156-165), no compositing; differentiable w.r.t. patch."""
    SH, SW = int(frame_size[0]), int(frame_size[1])
    dummy = patch.new_empty((1, 3, SH, SW))   # shape carrier only; never read in warp-only mode
    return _EotPaste.apply(dummy, _c(patch), _c(pmask), _c(coeffs), int(l_pad), int(t_pad), SH, SW,
                           N.PASTE_WARP_ONLY)


class _MaskedSqMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, mask):
        lib = N.lib()
        n = disp.numel()
        if mask is not None and mask.numel() != n:
            raise RuntimeError("masked_sq_mean: mask and disp sizes differ")
        part = torch.empty(lib.dmh_sq_mean_partials_size(n), device=disp.device, dtype=torch.float32)
        cost = torch.empty((), device=disp.device, dtype=torch.float32)
        N.check(lib.dmh_masked_sq_mean_fwd(N.ptr(disp), N.ptr(mask), n, N.ptr(part), N.ptr(cost), N.stream()))
        ctx.save_for_backward(disp, mask)
        return cost

    @staticmethod
    def backward(ctx, g):
        disp, mask = ctx.saved_tensors
        lib = N.lib()
        g_disp = torch.empty_like(disp)
        N.check(lib.dmh_masked_sq_mean_bwd(N.ptr(disp), N.ptr(mask), disp.numel(), N.ptr(_c(g.to(torch.float32))),
                                           N.ptr(g_disp), N.stream()))
        return g_disp, None


def masked_sq_mean(disp, mask=None):
    """mean((disp*mask)^2) == nn.MSELoss()(disp*mask, zeros) (phy_obj_atk.py:94)."""
    return _MaskedSqMean.apply(_c(disp), None if mask is None else _c(mask))


def _gt_depth_operands(disp, disp_gt, objmask, objdepth):
    """Checked operands of the --gt_depth term: (disp, disp_gt, mask tensor, mask batch stride, objdepth[B], B, HW).
    ``objmask`` is inputs[("color_objmask",0,0)] ([B,3,H,W] or [B,1,H,W]; channel 0 is read in place through its batch stride),
    ``objdepth`` inputs[("objdepth",0,0)] (one distance per sample, any shape with B elements)."""
    disp, disp_gt = _c(disp), _c(disp_gt)
    if disp.dim() != 4 or disp.shape[1] != 1 or disp_gt.shape != disp.shape:
        raise RuntimeError("gt_depth_mse: disp and disp_gt must both be [B,1,H,W]")
    B, _, H, W = disp.shape
    if objmask.dim() != 4 or objmask.shape[0] != B or tuple(objmask.shape[2:]) != (H, W):
        raise RuntimeError("gt_depth_mse: color_objmask must be [B,C,H,W] at the disparity's size")
    if objmask.stride(3) != 1 or objmask.stride(2) != W or objmask.stride(0) < H * W:
        objmask = objmask[:, :1].contiguous()      # e.g. the dataset's expand(-1, 3, -1, -1) view of a one-channel mask
    if objmask.dtype != torch.float32 or objmask.device != disp.device or not objmask.is_cuda:
        raise RuntimeError("gt_depth_mse: color_objmask must be float32 on the disparity's (ROCm) device")
    objdepth = _c(objdepth.reshape(-1))
    if objdepth.numel() != B:
        raise RuntimeError("gt_depth_mse: objdepth needs one distance per sample")
    return disp, disp_gt, objmask, int(objmask.stride(0)), objdepth, B, H * W


class _GtDepthMse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, disp_gt, objmask, objdepth, min_depth, max_depth):
        lib = N.lib()
        disp, disp_gt, objmask, bstride, objdepth, B, HW = _gt_depth_operands(disp, disp_gt, objmask, objdepth)
        part = torch.empty(lib.dmh_sq_mean_partials_size(B * HW), device=disp.device, dtype=torch.float32)
        cost = torch.empty((), device=disp.device, dtype=torch.float32)
        # channel 0 of the mask is read in place through its batch stride (layout validated by _gt_depth_operands)
        N.check(lib.dmh_gt_depth_mse_fwd(N.ptr(disp), N.ptr(disp_gt), C.c_void_p(objmask.data_ptr()), bstride, N.ptr(objdepth), B, HW,
                                         float(min_depth), float(max_depth), N.ptr(part), N.ptr(cost), N.stream()))
        ctx.save_for_backward(disp, disp_gt, objmask, objdepth)
        ctx.geo = (bstride, B, HW, float(min_depth), float(max_depth))
        return cost

    @staticmethod
    def backward(ctx, g):
        disp, disp_gt, objmask, objdepth = ctx.saved_tensors
        bstride, B, HW, min_depth, max_depth = ctx.geo
        g_disp = torch.empty_like(disp)
        N.check(N.lib().dmh_gt_depth_mse_bwd(N.ptr(disp), N.ptr(disp_gt), C.c_void_p(objmask.data_ptr()), bstride, N.ptr(objdepth), B, HW,
                                             min_depth, max_depth, N.ptr(_c(g.to(torch.float32))), N.ptr(g_disp), N.stream()))
        return g_disp, None, None, None, None, None


def gt_depth_mse(disp, disp_gt, objmask, objdepth, min_depth=0.1, max_depth=100.0):
    """--gt_depth supervised term (MD2/trainer.py:551-557): MSELoss(gt_depth, pred_depth) with
    pred / pseudo depth = clamp(disp_to_depth(.)[1] * 5.4, 1e-3, 80) and gt_depth = m objdepth + pseudo (1 - m), m = channel 0
    of color_objmask.  One fused pass forward, one backward (gradient w.r.t. ``disp`` only: disp_gt is the frozen teacher's)."""
    return _GtDepthMse.apply(disp, disp_gt.detach(), objmask, objdepth, min_depth, max_depth)


def avg_pyramid(x):
    """[F.avg_pool2d(x, 2), F.avg_pool2d(x, 4), F.avg_pool2d(x, 8)] of a [B,C,H,W] frame (H, W multiples of 8) in ONE launch,
    bit-identical to the three ATen calls: the colour pyramid inputs[("color", f, s)] of the GPU-side sample synthesis (no
    gradient: the frames are data)."""
    x = _c(x.detach())
    B, Cc, H, W = x.shape
    outs = [torch.empty((B, Cc, H >> s, W >> s), device=x.device, dtype=torch.float32) for s in (1, 2, 3)]
    N.check(_timed("avg_pyramid", lambda: N.lib().dmh_avg_pyramid(N.ptr(x), B * Cc, H, W, N.ptr(outs[0]), N.ptr(outs[1]),
                                                                  N.ptr(outs[2]), N.stream()), 4 * (x.numel() * 85 // 64)))
    return outs


def pgd_linf_step(x, x0, grad, alpha, eps, out=None):
    """x <- clamp(x0 + clamp(x + alpha*sign(grad) - x0, -eps, eps), 0, 1) (phy_obj_atk.py:98-101)."""
    x, x0, grad = _c(x.detach()), _c(x0), _c(grad)
    if x.shape != x0.shape or x.shape != grad.shape:
        raise RuntimeError("pgd_linf_step: shape mismatch")
    if out is None:
        out = torch.empty_like(x)
    N.check(N.lib().dmh_pgd_linf_step(N.ptr(x), N.ptr(x0), N.ptr(grad), float(alpha), float(eps), N.ptr(out),
                                      x.numel(), N.stream()))
    return out


class _L0Compose(torch.autograd.Function):
    @staticmethod
    def forward(ctx, obj, pos, neg, l0_clip, finalize):
        lib = N.lib()
        Cc, HW = obj.shape[1], obj.shape[2] * obj.shape[3]
        adv = torch.empty_like(obj)
        count = torch.zeros(1, device=obj.device, dtype=torch.int32)
        N.check(lib.dmh_l0_compose_fwd(N.ptr(obj), N.ptr(pos), N.ptr(neg), Cc, HW, l0_clip, int(finalize),
                                       N.ptr(adv), N.ptr(count), N.stream()))
        ctx.save_for_backward(obj, pos, neg)
        ctx.mark_non_differentiable(count)
        return adv, count

    @staticmethod
    def backward(ctx, g_adv, g_count):
        obj, pos, neg = ctx.saved_tensors
        lib = N.lib()
        Cc, HW = obj.shape[1], obj.shape[2] * obj.shape[3]
        g_pos, g_neg = torch.empty_like(pos), torch.empty_like(neg)
        N.check(lib.dmh_l0_compose_bwd(N.ptr(obj), N.ptr(pos), N.ptr(neg), N.ptr(_c(g_adv)), Cc, HW, N.ptr(g_pos),
                                       N.ptr(g_neg), 0, N.stream()))
        return None, g_pos, g_neg, None, None


def l0_compose(obj, pos, neg, l0_clip=1.0 / 255.0, finalize=False):
    """adv = clamp(obj + clamp(pos,0,1) - clamp(neg,0,1), 0, 1) and the L0 count of the thresholded
    pattern (phy_obj_atk_l0.py:94-99, 43-52; finalize=True applies :143-150).  Returns (adv, count[int32,1])."""
    if obj.dim() != 4 or obj.shape[0] != 1 or obj.shape != pos.shape or obj.shape != neg.shape:
        raise RuntimeError("l0_compose: obj/pos/neg must all be [1,C,H,W]")
    return _L0Compose.apply(_c(obj), _c(pos), _c(neg), float(l0_clip), bool(finalize))


class _L0MaskCost(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, neg):
        lib = N.lib()
        Cc, HW = pos.shape[1], pos.shape[2] * pos.shape[3]
        part = torch.empty(lib.dmh_l0_mask_partials_size(HW), device=pos.device, dtype=torch.float32)
        cost = torch.empty((), device=pos.device, dtype=torch.float32)
        N.check(lib.dmh_l0_mask_cost_fwd(N.ptr(pos), N.ptr(neg), Cc, HW, N.ptr(part), N.ptr(cost), N.stream()))
        ctx.save_for_backward(pos, neg)
        return cost

    @staticmethod
    def backward(ctx, g):
        pos, neg = ctx.saved_tensors
        lib = N.lib()
        Cc, HW = pos.shape[1], pos.shape[2] * pos.shape[3]
        g_pos, g_neg = torch.empty_like(pos), torch.empty_like(neg)
        N.check(lib.dmh_l0_mask_cost_bwd(N.ptr(pos), N.ptr(neg), Cc, HW, N.ptr(_c(g.to(torch.float32))), None,
                                         N.ptr(g_pos), N.ptr(g_neg), 0, N.stream()))
        return g_pos, g_neg


def l0_mask_cost(pos, neg):
    """mean(max_c(tanh(pos/10)/(2-1e-7)+0.5)) + same for neg (phy_obj_atk_l0.py:130-132)."""
    if pos.dim() != 4 or pos.shape[0] != 1 or pos.shape != neg.shape:
        raise RuntimeError("l0_mask_cost: pos/neg must be [1,C,H,W]")
    return _L0MaskCost.apply(_c(pos), _c(neg))


class _UpCatPad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, skip):
        lib = N.lib()
        B, C1, h, w = y.shape
        C2 = 0 if skip is None else skip.shape[1]
        if skip is not None and tuple(skip.shape) != (B, C2, 2 * h, 2 * w):
            raise RuntimeError("up_cat_pad: skip must be [B,C2,2h,2w], got %s for y %s" % (tuple(skip.shape), tuple(y.shape)))
        out = torch.empty((B, C1 + C2, 2 * h + 2, 2 * w + 2), device=y.device, dtype=torch.float32)
        nb = 4 * (y.numel() + (0 if skip is None else skip.numel()) + out.numel())
        N.check(_timed("up_cat_pad_fwd", lambda: lib.dmh_dec_up_cat_pad_fwd(N.ptr(y), N.ptr(skip), B, C1, C2, h, w,
                                                                           N.ptr(out), N.stream()), nb))
        ctx.save_for_backward(y)
        ctx.c2 = C2
        ctx.skip_grad = skip is not None and skip.requires_grad
        return out

    @staticmethod
    def backward(ctx, g_out):
        (y,) = ctx.saved_tensors
        lib = N.lib()
        B, C1, h, w = y.shape
        g_y = torch.empty_like(y)
        g_skip = torch.empty((B, ctx.c2, 2 * h, 2 * w), device=y.device, dtype=torch.float32) if ctx.skip_grad else None
        g_out = _c(g_out)
        nb = 4 * (2 * y.numel() + g_out.numel() + (0 if g_skip is None else g_skip.numel()))
        N.check(_timed("up_cat_pad_bwd", lambda: lib.dmh_dec_up_cat_pad_bwd(N.ptr(y), N.ptr(g_out), B, C1, ctx.c2, h, w,
                                                                           N.ptr(g_y), N.ptr(g_skip), N.stream()), nb))
        return g_y, g_skip


def up_cat_pad(y, skip=None):
    """pad1_reflect(cat(up2_nearest(ELU(y)), skip)) in one pass (MD2/networks/depth_decoder.py:54-60 + the
    ReflectionPad2d of the next Conv3x3)."""
    return _UpCatPad.apply(_c(y), None if skip is None else _c(skip))


class _EluPad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, apply_elu):
        lib = N.lib()
        B, Cc, H, W = z.shape
        out = torch.empty((B, Cc, H + 2, W + 2), device=z.device, dtype=torch.float32)
        N.check(_timed("elu_pad_fwd", lambda: lib.dmh_elu_pad_fwd(N.ptr(z), B, Cc, H, W, int(apply_elu), N.ptr(out),
                                                                 N.stream()), 4 * (z.numel() + out.numel())))
        ctx.save_for_backward(z)
        ctx.apply_elu = int(apply_elu)
        return out

    @staticmethod
    def backward(ctx, g_out):
        (z,) = ctx.saved_tensors
        lib = N.lib()
        B, Cc, H, W = z.shape
        g_z = torch.empty_like(z)
        g_out = _c(g_out)
        N.check(_timed("elu_pad_bwd", lambda: lib.dmh_elu_pad_bwd(N.ptr(z), N.ptr(g_out), B, Cc, H, W, ctx.apply_elu,
                                                                 N.ptr(g_z), N.stream()), 4 * (2 * z.numel() + g_out.numel())))
        return g_z, None


def elu_pad(z, apply_elu=True):
    """pad1_reflect(ELU(z)) in one pass (apply_elu=False: ReflectionPad2d(1) only)."""
    return _EluPad.apply(_c(z), bool(apply_elu))


class _BnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, shift, residual, relu):
        lib = N.lib()
        B, Cc = x.shape[0], x.shape[1]
        HW = x.numel() // (B * Cc)
        if scale.numel() != Cc or shift.numel() != Cc:
            raise RuntimeError("bn_act: scale/shift must have %d entries" % Cc)
        if residual is not None and residual.shape != x.shape:
            raise RuntimeError("bn_act: residual %s does not match x %s" % (tuple(residual.shape), tuple(x.shape)))
        out = torch.empty_like(x)
        nb = 4 * (x.numel() * (2 if residual is None else 3))
        N.check(_timed("bn_act_fwd", lambda: lib.dmh_bn_act_fwd(N.ptr(x), N.ptr(scale), N.ptr(shift), N.ptr(residual), B,
                                                               Cc, HW, int(relu), N.ptr(out), N.stream()), nb))
        ctx.save_for_backward(out if relu else None, scale)
        ctx.relu = int(relu)
        ctx.res_grad = residual is not None and residual.requires_grad
        ctx.x_grad = x.requires_grad
        return out

    @staticmethod
    def backward(ctx, g_out):
        out, scale = ctx.saved_tensors
        lib = N.lib()
        g_out = _c(g_out)
        B, Cc = g_out.shape[0], g_out.shape[1]
        HW = g_out.numel() // (B * Cc)
        g_x = torch.empty_like(g_out)
        g_res = torch.empty_like(g_out) if ctx.res_grad else None
        nb = 4 * g_out.numel() * (2 + (1 if ctx.relu else 0) + (1 if ctx.res_grad else 0))
        N.check(_timed("bn_act_bwd", lambda: lib.dmh_bn_act_bwd(N.ptr(out), N.ptr(g_out), N.ptr(scale), B, Cc, HW, ctx.relu,
                                                               N.ptr(g_x), N.ptr(g_res), N.stream()), nb))
        return g_x, None, None, g_res, None


def _reject_affine_grad(name, scale, shift):
    """scale / shift are constants of these ops (their backward returns None for them): refuse, loudly, a call that
    expects a gradient through them instead of silently dropping it."""
    if torch.is_grad_enabled() and not _wino_frozen and (scale.requires_grad or shift.requires_grad):
        raise RuntimeError("%s: scale/shift require grad, but this op treats them as constants (eval-mode BatchNorm "
                           "inside an attack); use the nn.BatchNorm2d module path, ops.frozen_weights() or detach()" % name)


def bn_act(x, scale, shift, residual=None, relu=True):
    """act(x * scale[c] + shift[c] (+ residual)) in one pass: BatchNorm2d in eval() mode folded to its per-channel
    affine, the BasicBlock's identity add and the ReLU (torchvision BasicBlock.forward as used by
    MD2/networks/resnet_encoder.py:85-98).  No gradient flows to scale/shift (eval-mode statistics and affine
    parameters are constants of the attack; phy_obj_atk.py:96 differentiates w.r.t. the patch only)."""
    _reject_affine_grad("bn_act", scale, shift)
    return _BnAct.apply(_c(x), _c(scale.detach()), _c(shift.detach()), None if residual is None else _c(residual),
                        bool(relu))


def _bn_train_stats(x, bn):
    """Batch statistics of x for the BatchNorm2d module `bn` in train mode: (scale, shift, save_mean, save_invstd);
    updates bn.running_mean / running_var / num_batches_tracked as nn.BatchNorm2d.forward does."""
    lib = N.lib()
    B, Cc = x.shape[0], x.shape[1]
    HW = x.numel() // (B * Cc)
    dev = x.device
    part = torch.empty(lib.dmh_bn_stats_partials_size(B, Cc, HW), device=dev, dtype=torch.float32)
    scale, shift, mean, invstd = (torch.empty(Cc, device=dev, dtype=torch.float32) for _ in range(4))
    if bn.momentum is None:
        raise RuntimeError("bn_act_train: cumulative-average BatchNorm (momentum=None) is not supported")
    track = bn.track_running_stats and bn.running_mean is not None
    nbt = bn.num_batches_tracked if (track and bn.num_batches_tracked is not None) else None
    if nbt is not None and not (nbt.is_cuda and nbt.dtype == torch.int64 and nbt.numel() == 1):
        raise RuntimeError("bn_act_train: num_batches_tracked must be a CUDA int64 scalar")
    N.check(_timed("bn_train_stats", lambda: lib.dmh_bn_train_stats_tracked(
        N.ptr(x), B, Cc, HW, N.ptr(None if bn.weight is None else _c(bn.weight.detach())),
        N.ptr(None if bn.bias is None else _c(bn.bias.detach())), float(bn.momentum), float(bn.eps),
        N.ptr(bn.running_mean if track else None), N.ptr(bn.running_var if track else None),
        None if nbt is None else C.c_void_p(nbt.data_ptr()), N.ptr(part), N.ptr(scale),
        N.ptr(shift), N.ptr(mean), N.ptr(invstd), N.stream()), 4 * x.numel()))
    return scale, shift, mean, invstd


def _bn_backward(g_pre, x, weight, bn, mean, invstd, eps, mask):
    """BatchNorm2d backward (train mode) from the saved batch statistics.  MIOpen's kernel when all three gradients
    are wanted (the train pass: 157 us average against 258 us for ATen's native kernel on these shapes)."""
    if all(mask) and weight is not None and hasattr(torch.ops.aten, "miopen_batch_norm_backward"):
        return torch.ops.aten.miopen_batch_norm_backward(x, g_pre, weight, bn.running_mean, bn.running_var, mean, invstd, eps)
    return torch.ops.aten.native_batch_norm_backward(g_pre, x, weight, bn.running_mean, bn.running_var, mean, invstd, True,
                                                     eps, mask)


def channel_sum(g):
    """sum of a contiguous fp32 CUDA [B, C, H, W] tensor over (0, 2, 3): a convolution's bias gradient, one streaming read."""
    lib = N.lib()
    B, Cc = g.shape[0], g.shape[1]
    HW = g.numel() // (B * Cc)
    part = torch.empty(lib.dmh_channel_sum_partials_size(B, Cc, HW), device=g.device, dtype=torch.float32)
    out = torch.empty(Cc, device=g.device, dtype=torch.float32)
    N.check(_timed("channel_sum", lambda: lib.dmh_channel_sum(N.ptr(g), B, Cc, HW, N.ptr(part), N.ptr(out), N.stream()),
                   4 * g.numel()))
    return out


BN_BWD_ENABLED = os.environ.get("DMH_BN_BWD", "1") != "0"      # A/B switch: 0 = K9 mask pass + MIOpen's backward


def _bn_backward_fused(g, x, out, weight, mean, invstd, want_pre):
    """K9 train-mode BatchNorm backward (dmh_bn_train_bwd): ReLU mask (``out`` = the saved ReLU output, or None) + the
    three gradients in three launches.  Returns (g_x, g_weight, g_bias, g_pre or None)."""
    lib = N.lib()
    B, Cc = g.shape[0], g.shape[1]
    HW = g.numel() // (B * Cc)
    ws = torch.empty(lib.dmh_bn_train_bwd_workspace_size(B, Cc, HW), device=g.device, dtype=torch.float32)
    gx = torch.empty_like(g)
    gw = torch.empty(Cc, device=g.device, dtype=torch.float32)
    gb = torch.empty(Cc, device=g.device, dtype=torch.float32)
    g_pre = torch.empty_like(g) if want_pre else None
    nb = 4 * g.numel() * ((6 if out is not None else 4) + 1 + (1 if want_pre else 0))
    N.check(_timed("bn_train_bwd", lambda: lib.dmh_bn_train_bwd(
        N.ptr(x), N.ptr(g), N.ptr(out), N.ptr(weight), N.ptr(mean), N.ptr(invstd), B, Cc, HW, N.ptr(ws), N.ptr(gx), N.ptr(gw),
        N.ptr(gb), N.ptr(g_pre), N.stream()), nb))
    return gx, gw, gb, g_pre


class _BnActTrain(torch.autograd.Function):
    """Train-mode BatchNorm2d -> (+ residual) -> ReLU: K9 statistics + one bn_act pass forward; backward = ReLU mask
    (one K9 pass) + aten::native_batch_norm_backward (MIOpen) with the saved batch statistics."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, relu, bn):
        lib = N.lib()
        scale, shift, mean, invstd = _bn_train_stats(x, bn)
        B, Cc = x.shape[0], x.shape[1]
        HW = x.numel() // (B * Cc)
        out = torch.empty_like(x)
        nb = 4 * (x.numel() * (2 if residual is None else 3))
        N.check(_timed("bn_act_fwd", lambda: lib.dmh_bn_act_fwd(N.ptr(x), N.ptr(scale), N.ptr(shift), N.ptr(residual), B, Cc,
                                                               HW, int(relu), N.ptr(out), N.stream()), nb))
        ctx.save_for_backward(x, weight, mean, invstd, out if relu else None)
        ctx.relu, ctx.eps, ctx.bn = bool(relu), float(bn.eps), bn
        ctx.res_grad = residual is not None and residual.requires_grad
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight, mean, invstd, out = ctx.saved_tensors
        lib = N.lib()
        g = _c(g)
        B, Cc = g.shape[0], g.shape[1]
        if BN_BWD_ENABLED and weight is not None and g.numel() < (1 << 31) and Cc <= 65535:
            gx, gw, gb, g_pre = _bn_backward_fused(g, x, out if ctx.relu else None, weight, mean, invstd,
                                                   ctx.res_grad and ctx.relu)
            if ctx.res_grad and not ctx.relu:
                g_pre = g
            return gx, gw, gb, (g_pre if ctx.res_grad else None), None, None
        if ctx.relu:
            ones = frozen_memo(("ones", Cc, g.device), lambda: torch.ones(Cc, device=g.device, dtype=torch.float32))
            g_pre = torch.empty_like(g)
            N.check(_timed("bn_act_bwd", lambda: lib.dmh_bn_act_bwd(N.ptr(out), N.ptr(g), N.ptr(ones), B, Cc,
                                                                   g.numel() // (B * Cc), 1, N.ptr(g_pre), None, N.stream()),
                           12 * g.numel()))
        else:
            g_pre = g
        gx, gw, gb = _bn_backward(g_pre, x, weight, ctx.bn, mean, invstd, ctx.eps,
                                  [ctx.needs_input_grad[0], weight is not None and ctx.needs_input_grad[1],
                                   weight is not None and ctx.needs_input_grad[2]])
        return gx, gw, gb, (g_pre if ctx.res_grad else None), None, None


def bn_act_train(bn, x, residual=None, relu=True):
    """act(bn(x) (+ residual)) for an nn.BatchNorm2d in train mode (batch statistics, running statistics updated):
    torchvision BasicBlock under model.train() (MD2/networks/resnet_encoder.py:85-98, the training pass)."""
    return _BnActTrain.apply(_c(x), bn.weight, bn.bias, None if residual is None else _c(residual), bool(relu), bn)


class _StemBnReluPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, shift):
        lib = N.lib()
        B, Cc, H, W = x.shape
        feat = torch.empty_like(x)
        pooled = torch.empty((B, Cc, H // 2, W // 2), device=x.device, dtype=torch.float32)
        arg = torch.empty((B, Cc, H // 2, W // 2), device=x.device, dtype=torch.uint8)
        nb = 4 * (2 * x.numel() + pooled.numel()) + arg.numel()
        N.check(_timed("stem_fwd", lambda: lib.dmh_stem_bn_relu_pool_fwd(N.ptr(x), N.ptr(scale), N.ptr(shift), B, Cc, H, W,
                                                                        N.ptr(feat), N.ptr(pooled), N.ptr(arg),
                                                                        N.stream()), nb))
        ctx.save_for_backward(feat, arg, scale)
        ctx.mark_non_differentiable(arg)
        ctx.set_materialize_grads(False)    # an unused output hands None to backward: the kernel skips that read
        return feat, pooled, arg

    @staticmethod
    def backward(ctx, g_feat, g_pooled, _g_arg):
        feat, arg, scale = ctx.saved_tensors
        if g_feat is None and g_pooled is None:
            return None, None, None
        lib = N.lib()
        B, Cc, H, W = feat.shape
        g_feat = None if g_feat is None else _c(g_feat)
        g_pooled = None if g_pooled is None else _c(g_pooled)
        g_x = torch.empty_like(feat)
        nb = 4 * (2 * feat.numel() + (0 if g_feat is None else feat.numel()) +
                  (0 if g_pooled is None else g_pooled.numel())) + arg.numel()
        N.check(_timed("stem_bwd", lambda: lib.dmh_stem_bn_relu_pool_bwd(N.ptr(feat), N.ptr(arg), N.ptr(g_feat),
                                                                        N.ptr(g_pooled), N.ptr(scale), B, Cc, H, W,
                                                                        N.ptr(g_x), N.stream()), nb))
        return g_x, None, None


class _StemTrain(torch.autograd.Function):
    """Train-mode stem: BatchNorm2d (batch statistics) -> ReLU -> MaxPool2d(3, 2, 1) with the K9 kernels; backward =
    max-pool adjoint + skip gradient + ReLU mask in one K9 pass, then aten::native_batch_norm_backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn):
        lib = N.lib()
        scale, shift, mean, invstd = _bn_train_stats(x, bn)
        B, Cc, H, W = x.shape
        feat = torch.empty_like(x)
        pooled = torch.empty((B, Cc, H // 2, W // 2), device=x.device, dtype=torch.float32)
        arg = torch.empty((B, Cc, H // 2, W // 2), device=x.device, dtype=torch.uint8)
        nb = 4 * (2 * x.numel() + pooled.numel()) + arg.numel()
        N.check(_timed("stem_fwd", lambda: lib.dmh_stem_bn_relu_pool_fwd(N.ptr(x), N.ptr(scale), N.ptr(shift), B, Cc, H, W,
                                                                        N.ptr(feat), N.ptr(pooled), N.ptr(arg),
                                                                        N.stream()), nb))
        ctx.save_for_backward(x, weight, mean, invstd, feat, arg)
        ctx.eps, ctx.bn = float(bn.eps), bn
        ctx.mark_non_differentiable(arg)
        ctx.set_materialize_grads(False)
        return feat, pooled, arg

    @staticmethod
    def backward(ctx, g_feat, g_pooled, _g_arg):
        x, weight, mean, invstd, feat, arg = ctx.saved_tensors
        if g_feat is None and g_pooled is None:
            return None, None, None, None
        lib = N.lib()
        B, Cc, H, W = feat.shape
        g_feat = None if g_feat is None else _c(g_feat)
        g_pooled = None if g_pooled is None else _c(g_pooled)
        ones = frozen_memo(("ones", Cc, feat.device), lambda: torch.ones(Cc, device=feat.device, dtype=torch.float32))
        g_pre = torch.empty_like(feat)
        N.check(_timed("stem_bwd", lambda: lib.dmh_stem_bn_relu_pool_bwd(N.ptr(feat), N.ptr(arg), N.ptr(g_feat),
                                                                        N.ptr(g_pooled), N.ptr(ones), B, Cc, H, W,
                                                                        N.ptr(g_pre), N.stream()), 12 * feat.numel()))
        if BN_BWD_ENABLED and weight is not None and g_pre.numel() < (1 << 31) and Cc <= 65535:
            gx, gw, gb, _ = _bn_backward_fused(g_pre, x, None, weight, mean, invstd, False)
            return gx, gw, gb, None
        gx, gw, gb = _bn_backward(g_pre, x, weight, ctx.bn, mean, invstd, ctx.eps,
                                  [ctx.needs_input_grad[0], weight is not None and ctx.needs_input_grad[1],
                                   weight is not None and ctx.needs_input_grad[2]])
        return gx, gw, gb, None


def stem_bn_relu_pool_train(bn, x):
    """(relu(bn(x)), maxpool3x3/2(relu(bn(x)))) for an nn.BatchNorm2d in train mode; H and W even."""
    feat, pooled, _ = _StemTrain.apply(_c(x), bn.weight, bn.bias, bn)
    return feat, pooled


def stem_bn_relu_pool(x, scale, shift):
    """(ReLU(BN_eval(x)), MaxPool2d(3, 2, 1) of it) in one pass -- the encoder stem of
    MD2/networks/resnet_encoder.py:88-91 (features[0] and the input of layer1).  H and W must be even."""
    _reject_affine_grad("stem_bn_relu_pool", scale, shift)
    feat, pooled, _ = _StemBnReluPool.apply(_c(x), _c(scale.detach()), _c(shift.detach()))
    return feat, pooled


# ---------------------------------------------------------------------------------------------------------------
# K10: 3x3 stride-1 convolution, Winograd F(2x2,3x3) on the fp32 matrix cores (forward and backward-data); weight gradients:
# K13 / K16 / K18 by channel counts.  Dispatch is by shape: the kernel wins where a launch fills the chip with >= 64 output
# channels (tools/wino_bench.py, profiles/); everything else is the ATen/MIOpen convolution.
# ---------------------------------------------------------------------------------------------------------------
WINO_ENABLED = os.environ.get("DMH_WINO", "1") != "0"
WRW_ENABLED = os.environ.get("DMH_WRW", "1") != "0"     # K18 / K20 / K21 (weight gradients); 0: MIOpen (A/B switch)
_wino_frozen = 0
_wino_cache = {}


@contextlib.contextmanager
def frozen_weights():
    """Scope in which the network parameters are constants (an attack: torchattacks/attack.py brackets it with
    model.eval() and differentiates w.r.t. the perturbation only -- phy_obj_atk.py:96, phy_obj_atk_l0.py:134,
    pgd_depth.py:72).  Inside, the Winograd-transformed filters are computed once per weight tensor instead of once
    per call (the cache is dropped on exit) and ops.conv3x3 produces no weight/bias gradients: a custom autograd
    Function cannot see that torch.autograd.grad() was asked for the input gradient only, and the weight-gradient
    convolutions would otherwise run in every attack step just to be thrown away."""
    global _wino_frozen
    _wino_frozen += 1
    try:
        yield
    finally:
        _wino_frozen -= 1
        if _wino_frozen == 0:
            _wino_cache.clear()


def weights_frozen():
    """True inside a frozen_weights() scope (network parameters are constants there)."""
    return _wino_frozen > 0


def frozen_memo(key, fn):
    """fn() memoised for the lifetime of the enclosing frozen_weights() scope (parameters are constants there);
    outside a scope fn() is simply evaluated."""
    if not _wino_frozen:
        return fn()
    if key not in _wino_cache:
        _wino_cache[key] = fn()
    return _wino_cache[key]


def _wino_filter(weight, backward, scale=None):
    """Winograd-transformed filter of the forward (backward=False) or backward-data pass; `scale` folds a per-channel
    factor of the forward OUTPUT (an eval-mode BatchNorm) into it."""
    lib = N.lib()
    K, Cc = weight.shape[0], weight.shape[1]
    key = (weight.data_ptr(), weight._version, bool(backward), None if scale is None else scale.data_ptr())
    if _wino_frozen and key in _wino_cache:
        return _wino_cache[key]
    if scale is None and not _wino_frozen:
        ready = _wino_ready.get((key[0], key[2]))      # transformed ahead of its use by wino_prefetch()
        if ready is not None and ready[0] == key[1] and ready[1].device == weight.device:
            return ready[1]
    n_out, n_in = (Cc, K) if backward else (K, Cc)
    U = torch.empty(lib.dmh_wino_weight_size(n_out, n_in), device=weight.device, dtype=torch.float32)
    N.check(lib.dmh_wino_weight_transform_scaled(N.ptr(_c(weight.detach())), K, Cc, int(backward), N.ptr(scale), N.ptr(U),
                                                 N.stream()))
    if _wino_frozen:
        _wino_cache[key] = U
    return U


WINO_PREFETCH = os.environ.get("DMH_WINO_PREFETCH", "1") != "0"     # A/B switch: 0 = every filter transformed at its first use
_wino_ready = {}        # (weight pointer, backward) -> (weight version, U, weight): the filters of ONE pass outside a
                        # frozen_weights() scope (the train pass: forward forms now, backward-data forms for its backward); an
                        # entry keeps its weight tensor alive, so the pointer cannot come to name another tensor's data


def wino_prefetch(jobs, fresh=False):
    """Transform many K10 filters in ONE launch (dmh_wino_weight_transform_batch) ahead of their use: ``jobs`` = iterable of
    (weight [K, C, 3, 3], backward, scale or None) -- what _wino_filter() will be asked for.  A step needs every filter in its
    forward and its backward-data form, for the attack's frozen weights (BatchNorm scale folded in) and again for the train
    pass: 76 launches of 5-20 us when each is made at its first use.  Forms the kernel does not take (input channels of the pass
    not a multiple of 8) are skipped; whatever is not prefetched is still transformed on demand.  Inside frozen_weights() the
    results join that scope's cache (scaled forms included); outside it only unscaled forms are taken, into a table that the
    first prefetch of the next pass (``fresh``: the encoder's) empties.  Returns the number of filters transformed."""
    if not (WINO_ENABLED and WINO_PREFETCH):
        return 0
    lib = N.lib()
    todo, keep = [], []
    if fresh and not _wino_frozen:
        _wino_ready.clear()
    for weight, backward, scale in jobs:
        if not (weight.is_cuda and weight.dtype == torch.float32 and weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3)):
            continue
        K, Cc = weight.shape[0], weight.shape[1]
        n_out, n_in = (Cc, K) if backward else (K, Cc)
        if n_in % 8 or n_in < 24 or n_out < 64:        # not a K10 shape in this direction (ops._wino_ok)
            continue
        if scale is not None and not _wino_frozen:     # a scale is a temporary: only a frozen scope's cache may key on it
            continue
        sc = None if scale is None else _c(scale.detach())
        key = (weight.data_ptr(), weight._version, bool(backward), None if sc is None else sc.data_ptr())
        if _wino_frozen and key in _wino_cache:
            continue
        # (outside a frozen scope a job is always done: an entry of an earlier call may predate a change of the weight that
        #  did not bump its version counter -- p.data.copy_(...))
        U = torch.empty(lib.dmh_wino_weight_size(n_out, n_in), device=weight.device, dtype=torch.float32)
        w = _c(weight.detach())
        keep.append((w, sc))
        todo.append((N.ptr(w), N.ptr(sc), N.ptr(U), K, Cc, int(bool(backward))))
        if _wino_frozen:
            _wino_cache[key] = U
        else:
            _wino_ready[(key[0], key[2])] = (key[1], U, weight)
    if todo:
        arr = (N.WinoWtJob * len(todo))()
        for j, (w, sc, U, K, Cc, bw) in enumerate(todo):
            arr[j].w, arr[j].scale, arr[j].U, arr[j].K, arr[j].C, arr[j].backward = w, sc, U, K, Cc, bw
        N.check(lib.dmh_wino_weight_transform_batch(arr, len(todo), N.stream()))
    return len(todo)


WINO_SK = os.environ.get("DMH_WINO_SK", "1") != "0"       # A/B switch: stream-K decomposition of the plain K10 launches
# work items below which MIOpen is level or ahead (A/B switch).  200 until round 5: measured with whole items only.  With the
# stream-K forms a launch of 64 items (or 512 (item, chunk) units) already beats the library's fixed costs: the strong-scaling
# share (train batch 4, 2 attack scenes) 50.2 -> 38.5 ms per step at 75 / 50 / 35 alike, batch 4 + 12 scenes and the headline
# batch unchanged (tools/ab_min_items.sh)
_WINO_MIN_ITEMS = int(os.environ.get("DMH_WINO_MIN_ITEMS", "64"))
_WINO_MIN_FILL = float(os.environ.get("DMH_WINO_MIN_FILL", "0.5"))      # real tiles / tiles of the regions below which MIOpen is ahead (A/B switch)


def _wino_ok(B, n_in, n_out, Ho, Wo, allow_split=True, allow_sk=False):
    """Shapes the Winograd-MFMA kernel takes: channel counts it tiles without waste and enough 64-channel x 64-tile
    work items to fill the 256 CUs (measured crossover, tools/wino_bench.py)."""
    if not WINO_ENABLED or n_in % 8 or n_in < 24 or n_out < 64 or Ho % 2 or Wo % 2 or Ho < 2 or Wo < 2:
        return False
    if B * n_in * (Ho + 2) * (Wo + 2) >= (1 << 30):     # 32-bit byte offsets of the kernel's buffer loads: ATen beyond
        return False
    if n_in < 64 and n_out % 64:        # few chunks per item and a half-empty channel group: MIOpen is level or ahead
        return False
    ht, wt = Ho // 2, Wo // 2
    narrow = wt % 32 != 0 and (wt <= 16 or (-wt) % 16 < (-wt) % 32)
    if narrow:      # 4x16 regions; rows of tiles flattened over the batch when per-image regions would waste >= 1/5
        flat = 5 * ht <= 4 * (-(-ht // 4) * 4)
        rows = -(-(B * ht) // 4) * 4 if flat else B * -(-ht // 4) * 4
        cols = -(-wt // 16) * 16
    else:
        rows, cols = B * -(-ht // 2) * 2, -(-wt // 32) * 32
    if B * ht * wt < _WINO_MIN_FILL * rows * cols:     # ragged image: too many empty tiles (e.g. 17 tile columns in a 32-wide region)
        return False
    regions = (rows // (4 if narrow else 2)) * (cols // (16 if narrow else 32))
    regions *= -(-n_out // 64)
    nch = n_in // 8
    split = 2 if (allow_split and regions < 192 and nch % 2 == 0 and nch >= 6) else 1      # mirrors launch_split()
    if WINO_SK and (allow_split or allow_sk) and regions * nch >= 8 * _WINO_MIN_ITEMS:
        # enough (item, chunk) units for the stream-K form -- IF the library takes it for this shape: its own decision is asked
        # (dmh_wino_conv3x3_plan: cost model, workspace size, the fused epilogue's few-regions rule), not mirrored here
        try:
            ws_floats = _sk_ws_floats(torch.cuda.current_device()) if torch.cuda.is_available() else 2 * 256 * _SK_SLOT_FLOATS
        except Exception:
            ws_floats = 2 * 256 * _SK_SLOT_FLOATS
        plan = N.lib().dmh_wino_conv3x3_plan(B, n_in, n_out, Ho, Wo, 1, 0 if allow_split else 1, ws_floats)
        if plan >= 0 and (plan & 1):
            return True
    return regions * split >= (_WINO_MIN_ITEMS if WINO_SK else max(_WINO_MIN_ITEMS, 200))     # whole items only: round 4's threshold


def _wrw_ok(x, g, K, Cc):
    """Shapes of K18 (Winograd-domain weight gradient): both channel counts multiples of 64 (64 x 64 blocks), or 32 k output
    channels with 96 k' / 64 k' input channels (32 x 96 / 32 x 64 blocks: upconv(1,1), upconv(1,0)); even output size, enough
    tile chunks (8 tiles) to give every workgroup of a (k-block, c-block) pair a few, tensors below 4 GB."""
    if not WINO_ENABLED or not WRW_ENABLED or g.shape[2] % 2 or g.shape[3] % 2:
        return False
    if K % 64 == 0 and Cc % 64 == 0:
        kch, cch = 64, 64
    elif K % 32 == 0 and Cc % 96 == 0:
        kch, cch = 32, 96
    elif K % 32 == 0 and Cc % 64 == 0:
        kch, cch = 32, 64
    else:
        return False
    pad = (g.shape[2] + 2 - x.shape[2]) // 2
    if pad not in (0, 1) or (pad == 1 and x.shape[3] % 16):
        return False
    if x.numel() * 4 > 0xFFFFFF00 or g.numel() >= (1 << 30):      # the kernel's own limits (DMH_REQUIRE in csrc/wino_wrw.hip)
        return False
    chunks = g.shape[0] * (g.shape[2] // 2) * -(-(g.shape[3] // 2) // 8)
    return chunks * (K // kch) * (Cc // cch) >= 1024


def _wino32_ok(B, n_in, n_out, Ho, Wo):
    """Shapes of K17, the 32-output-channel form of K10 (work item 32 channels x 4 x 32 tiles): output channels a multiple
    of 32 that K10's 64-channel items would half-fill, enough items to cover the chip and few empty tiles."""
    if (not WINO_ENABLED or n_in % 8 or n_in < 32 or n_out % 32 or n_out % 64 == 0 or n_out > 96 or Ho % 2 or Wo % 2
            or B * n_in * (Ho + 2) * (Wo + 2) >= (1 << 30)):
        return False        # (32 -> 96, the backward-data pass of upconv(1,1): 509 us against MIOpen's 566, tools/wino32_bench.py)
    ht, wt = Ho // 2, Wo // 2
    rows, cols = -(-ht // 4) * 4, -(-wt // 32) * 32
    if ht * wt < 0.8 * rows * cols:
        return False
    return B * (rows // 4) * (cols // 32) * (n_out // 32) >= 400


def _wino32_filter(weight, backward):
    lib = N.lib()
    K, Cc = weight.shape[0], weight.shape[1]
    key = (weight.data_ptr(), weight._version, bool(backward), "k17")
    if _wino_frozen and key in _wino_cache:
        return _wino_cache[key]
    n_out, n_in = (Cc, K) if backward else (K, Cc)
    U = torch.empty(lib.dmh_wino32_weight_size(n_out, n_in), device=weight.device, dtype=torch.float32)
    N.check(lib.dmh_wino32_weight_transform(N.ptr(_c(weight.detach())), K, Cc, int(backward), N.ptr(U), N.stream()))
    if _wino_frozen:
        _wino_cache[key] = U
    return U


def _wino32_conv(x, U, bias, K, pad):
    lib = N.lib()
    B, Cc, H, W = x.shape
    y = torch.empty((B, K, H + 2 * pad - 2, W + 2 * pad - 2), device=x.device, dtype=torch.float32)
    nb = 4 * (x.numel() + y.numel()) + 4 * U.numel()
    ws = _sk_workspace(x.device) if WINO_SK else None
    N.check(_timed("wino32_conv3x3", lambda: lib.dmh_wino32_conv3x3_ws(
        N.ptr(x), N.ptr(U), N.ptr(bias), B, Cc, K, H, W, pad, N.ptr(y), N.ptr(ws), 0 if ws is None else ws.numel(), N.stream()),
        nb, 18 * Cc * y.numel()))
    return y


_sk_ws = {}         # (device, stream) -> workspace of the stream-K launches (caller-owned: the library keeps nothing)
_SK_SLOT_FLOATS = 16384         # one partial work item: 16 output channels x 256 threads x float4


def _sk_ws_floats(device):
    """2 slots per workgroup, one workgroup per CU (MI355X: 256 CUs = 32 MB): sized from the device, not assumed."""
    return 2 * torch.cuda.get_device_properties(device).multi_processor_count * _SK_SLOT_FLOATS


def _sk_workspace(device):
    """The library decides per launch (dmh_wino_conv3x3_ws / dmh_wino32_conv3x3_ws): stream-K where whole work items would leave
    the chip idle for part of a round.  ONE buffer per device AND stream -- the launches of a stream are ordered, and the fix-up
    kernel that reads the partial items is enqueued right behind the kernel that wrote them; two streams that run convolutions
    side by side must not share the slots."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    ws = _sk_ws.get(key)
    if ws is None:
        ws = _sk_ws[key] = torch.empty(_sk_ws_floats(device), device=device, dtype=torch.float32)
    return ws


def _k10_act(lib, x, U, bias, res, relu, B, Cc, K, H, W, pad, y, stream):
    """dmh_wino_conv3x3_act with the stream-K workspace of the current device (the library takes the decomposed form only for
    launches of few tile regions: layer4 at the attack batch)."""
    if not WINO_SK:
        return lib.dmh_wino_conv3x3_act(x, U, bias, res, relu, B, Cc, K, H, W, pad, y, stream)
    ws = _sk_workspace(torch.device("cuda", torch.cuda.current_device()))
    return lib.dmh_wino_conv3x3_act_ws(x, U, bias, res, relu, B, Cc, K, H, W, pad, y, N.ptr(ws), ws.numel(), stream)


def _wino_conv(x, U, bias, K, pad):
    lib = N.lib()
    B, Cc, H, W = x.shape
    y = torch.empty((B, K, H + 2 * pad - 2, W + 2 * pad - 2), device=x.device, dtype=torch.float32)
    nb = 4 * (x.numel() + y.numel()) + 4 * U.numel()
    ws = _sk_workspace(x.device) if WINO_SK else None
    N.check(_timed("wino_conv3x3", lambda: lib.dmh_wino_conv3x3_ws(
        N.ptr(x), N.ptr(U), N.ptr(bias), B, Cc, K, H, W, pad, N.ptr(y), N.ptr(ws), 0 if ws is None else ws.numel(), N.stream()),
        nb, 18 * Cc * y.numel()))
    return y


def _small_ok(n_in, n_out):
    """Channel counts of the K11 direct-MFMA kernel (last decoder stage, disparity heads)."""
    return WINO_ENABLED and ((n_in in (1, 2, 3, 4, 16) and n_out <= 32) or (n_in == 32 and n_out <= 16))


def _small_conv(x, weight, bias, pad, backward):
    lib = N.lib()
    if x.numel() >= (1 << 30):      # beyond the 32-bit byte offsets of the kernel's buffer loads: ATen, as for K15 / K18
        if backward:
            return torch.nn.functional.conv_transpose2d(x, weight, None, 1, 2 - pad)
        return torch.conv2d(x, weight, bias, 1, pad)
    B, _, H, W = x.shape
    Kw, Cw = weight.shape[0], weight.shape[1]
    n_out = Cw if backward else Kw
    y = torch.empty((B, n_out, H + 2 * pad - 2, W + 2 * pad - 2), device=x.device, dtype=torch.float32)
    nb = 4 * (x.numel() + y.numel())
    N.check(_timed("conv3x3_small", lambda: lib.dmh_conv3x3_small(N.ptr(x), N.ptr(_c(weight.detach())), N.ptr(bias), B, Kw,
                                                                 Cw, H, W, pad, int(backward), N.ptr(y), N.stream()), nb,
                   18 * x.shape[1] * y.numel()))
    return y


class _Conv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, pad):
        B, Cc, H, W = x.shape
        K = weight.shape[0]
        ctx.pad = pad
        ctx.has_bias = bias is not None
        ctx.params_const = _wino_frozen > 0
        ctx.save_for_backward(x, weight)
        if _wino_ok(B, Cc, K, H + 2 * pad - 2, W + 2 * pad - 2):
            return _wino_conv(x, _wino_filter(weight, False), None if bias is None else _c(bias.detach()), K, pad)
        if _wino32_ok(B, Cc, K, H + 2 * pad - 2, W + 2 * pad - 2):
            return _wino32_conv(x, _wino32_filter(weight, False), None if bias is None else _c(bias.detach()), K, pad)
        if WINO_ENABLED and K == 1 and ((pad == 0 and Cc % 4 == 0) or (      # disparity head: K13 (strip kernel at pad 0)
                Cc % 16 == 0 and Cc >= 32 and B * -(-(H + 2 * pad - 2) // 8) * -(-(W + 2 * pad - 2) // 64) >= 512)):
            lib = N.lib()
            y = torch.empty((B, 1, H + 2 * pad - 2, W + 2 * pad - 2), device=x.device, dtype=torch.float32)
            N.check(_timed("conv3x3_head", lambda: lib.dmh_conv3x3_head(
                N.ptr(x), N.ptr(_c(weight.detach())), N.ptr(None if bias is None else _c(bias.detach())), B, Cc, H, W, pad,
                N.ptr(y), N.stream()), 4 * (x.numel() + y.numel())))
            return y
        if _small_ok(Cc, K):
            return _small_conv(x, weight, None if bias is None else _c(bias.detach()), pad, False)
        return torch.conv2d(x, weight, bias, 1, pad)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        B, Cc, H, W = x.shape
        K = weight.shape[0]
        pad = ctx.pad
        g = _c(g)
        need_x = ctx.needs_input_grad[0]
        need_w = ctx.needs_input_grad[1] and not ctx.params_const
        need_b = ctx.has_bias and ctx.needs_input_grad[2] and not ctx.params_const
        g_x = g_w = g_b = None
        if need_x and _wino_ok(B, K, Cc, H, W):
            # backward-data = the same convolution on g with the flipped/transposed filter and pad' = 2 - pad
            g_x = _wino_conv(g, _wino_filter(weight, True), None, Cc, 2 - pad)
            need_x = False
        elif need_x and _wino32_ok(B, K, Cc, H, W):
            g_x = _wino32_conv(g, _wino32_filter(weight, True), None, Cc, 2 - pad)
            need_x = False
        elif need_x and K == 1 and pad == 0 and Cc % 4 == 0 and Cc <= 64 and H >= 3 and W >= 3 and WINO_ENABLED:
            lib = N.lib()           # disparity head: one gradient plane in registers, C planes streamed out
            g_x = torch.empty_like(x)
            N.check(_timed("conv3x3_head_bwd", lambda: lib.dmh_conv3x3_head_bwd_data(
                N.ptr(g), N.ptr(_c(weight.detach())), B, Cc, H, W, N.ptr(g_x), N.stream()), 4 * (g_x.numel() + g.numel())))
            need_x = False
        elif need_x and _small_ok(K, Cc):
            g_x = _small_conv(g, weight, None, 2 - pad, True)
            need_x = False
        if need_w and K == 1 and WINO_ENABLED and Cc * H * W < (1 << 29):
            # disparity head (one output channel): K13's strip reduction instead of MIOpen's 1-row implicit GEMM
            lib = N.lib()
            part = torch.empty(lib.dmh_conv3x3_head_wrw_partials_size(B, Cc, H, W, pad), device=g.device, dtype=torch.float32)
            g_w = torch.empty_like(weight)
            g_b = torch.empty(1, device=g.device, dtype=torch.float32) if need_b else None
            N.check(_timed("conv3x3_head_wrw", lambda: lib.dmh_conv3x3_head_wrw(
                N.ptr(x), N.ptr(g), B, Cc, H, W, pad, N.ptr(part), N.ptr(g_w), N.ptr(g_b), N.stream()),
                4 * (x.numel() + g.numel())))
            need_w = need_b = False
        if need_w and K == 16 and Cc in (16, 32) and WINO_ENABLED and Cc * H * W < (1 << 28):
            # last decoder stage: K16 (pixel axis on the MFMA, persistent workgroups) instead of MIOpen's implicit GEMM
            lib = N.lib()
            part = torch.empty(lib.dmh_conv3x3_small_wrw_partials_size(Cc), device=g.device, dtype=torch.float32)
            g_w = torch.empty_like(weight)
            g_b = torch.empty(16, device=g.device, dtype=torch.float32) if need_b else None
            N.check(_timed("conv3x3_small_wrw", lambda: lib.dmh_conv3x3_small_wrw(
                N.ptr(x), N.ptr(g), B, Cc, H, W, pad, N.ptr(part), N.ptr(g_w), N.ptr(g_b), N.stream()),
                4 * (x.numel() + g.numel()), 2 * 9 * 16 * Cc * g.numel() // 16))
            need_w = need_b = False
        if need_w and _wrw_ok(x, g, K, Cc):
            # >= 64 channels on both sides: K18, the weight gradient in the Winograd domain on the fp32 MFMA (MIOpen: NHWC
            # implicit GEMMs between layout transposes, with atomics); the bias gradient is a plain reduction of g
            lib = N.lib()
            ws = torch.empty(lib.dmh_wino_wrw_workspace_size(B, Cc, K, H, W, pad), device=g.device, dtype=torch.float32)
            g_w = torch.empty_like(weight)
            N.check(_timed("wino_wrw", lambda: lib.dmh_wino_wrw(N.ptr(x), N.ptr(g), B, Cc, K, H, W, pad, N.ptr(ws), N.ptr(g_w),
                                                                N.stream()), 4 * (x.numel() + g.numel()), 18 * Cc * g.numel()))
            if need_b:
                g_b = channel_sum(g) if K <= 65535 else g.sum((0, 2, 3))
            need_w = need_b = False
        if need_x or need_w or need_b:
            r = torch.ops.aten.convolution_backward(g, x, weight, [K] if ctx.has_bias else None, [1, 1], [pad, pad], [1, 1],
                                                    False, [0, 0], 1, [need_x, need_w, need_b])
            g_x = r[0] if need_x else g_x
            g_w = r[1] if need_w else g_w
            g_b = r[2] if need_b else g_b
        return g_x, g_w, g_b, None


class _ConvBnAct(torch.autograd.Function):
    """conv3x3 -> eval-mode BatchNorm -> (+ residual) -> ReLU as ONE K10 launch (scale folded into the filter, shift as
    the bias, residual and ReLU in the output transform).  Backward: ReLU mask (one K9 pass), then the backward-data
    K10 launch on the filter with the scale folded in."""

    @staticmethod
    def forward(ctx, x, weight, scale, shift, residual, relu, pad):
        K = weight.shape[0]
        lib = N.lib()
        B, Cc, H, W = x.shape
        y = torch.empty((B, K, H + 2 * pad - 2, W + 2 * pad - 2), device=x.device, dtype=torch.float32)
        U = _wino_filter(weight, False, scale)
        nb = 4 * (x.numel() + y.numel() * (1 if residual is None else 2)) + 4 * U.numel()
        N.check(_timed("wino_conv3x3", lambda: _k10_act(lib, N.ptr(x), N.ptr(U), N.ptr(shift), N.ptr(residual),
                                                                       int(relu), B, Cc, K, H, W, pad, N.ptr(y),
                                                                       N.stream()), nb, 18 * Cc * y.numel()))
        ctx.save_for_backward(x, weight, scale, y if relu else None)
        ctx.pad, ctx.relu = pad, bool(relu)
        ctx.res_grad = residual is not None and residual.requires_grad
        ctx.params_const = _wino_frozen > 0
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, scale, y = ctx.saved_tensors
        lib = N.lib()
        B, Cc, H, W = x.shape
        K = weight.shape[0]
        g = _c(g)
        if ctx.relu:        # g_pre = g * [y > 0] (K9 kernel with a unit scale); it is also the residual's gradient
            ones = frozen_memo(("ones", K, g.device), lambda: torch.ones(K, device=g.device, dtype=torch.float32))
            g_pre = torch.empty_like(g)
            N.check(_timed("bn_act_bwd", lambda: lib.dmh_bn_act_bwd(N.ptr(y), N.ptr(g), N.ptr(ones), B, K, g.numel() // (B * K),
                                                                   1, N.ptr(g_pre), None, N.stream()), 12 * g.numel()))
        else:
            g_pre = g
        g_x = g_w = None
        need_w = ctx.needs_input_grad[1] and not ctx.params_const
        if ctx.needs_input_grad[0]:
            if _wino_ok(B, K, Cc, H, W):
                g_x = _wino_conv(g_pre, _wino_filter(weight, True, scale), None, Cc, 2 - ctx.pad)
            else:
                g_conv = g_pre * scale.view(1, -1, 1, 1)
                g_x = torch.ops.aten.convolution_backward(g_conv, x, weight, None, [1, 1], [ctx.pad] * 2, [1, 1], False,
                                                          [0, 0], 1, [True, False, False])[0]
        if need_w:
            g_conv = g_pre * scale.view(1, -1, 1, 1)
            g_w = torch.ops.aten.convolution_backward(g_conv, x, weight, None, [1, 1], [ctx.pad] * 2, [1, 1], False, [0, 0],
                                                      1, [False, True, False])[1]
        return g_x, g_w, None, None, (g_pre if ctx.res_grad else None), None, None


def conv3x3_bn_act(x, weight, scale, shift, residual=None, relu=True, padding=1):
    """act(BatchNorm_eval(conv3x3(x, weight)) (+ residual)) with BatchNorm given as its per-channel (scale, shift): the
    BasicBlock pattern of the encoder in eval() (MD2/networks/resnet_encoder.py:85-98).  One K10 launch where the shape
    fills the chip, otherwise conv3x3 followed by the K9 bn_act pass.  No gradient flows to scale / shift."""
    B, Cc, H, W = x.shape
    _reject_affine_grad("conv3x3_bn_act", scale, shift)
    if x.is_cuda and _wino_ok(B, Cc, weight.shape[0], H + 2 * padding - 2, W + 2 * padding - 2, allow_split=False, allow_sk=True):
        return _ConvBnAct.apply(_c(x), weight, _c(scale.detach()), _c(shift.detach()),
                                None if residual is None else _c(residual), bool(relu), int(padding))
    return bn_act(conv3x3(x, weight, None, padding), scale, shift, residual, relu)


def conv3x3(x, weight, bias=None, padding=1):
    """nn.Conv2d(C, K, 3, stride=1, padding=padding) with zero padding 0, 1 or 2 -- the 3x3 convolutions of the ResNet
    encoder (torchvision BasicBlock under MD2/networks/resnet_encoder.py:85-98) and of the depth decoder
    (MD2/layers.py:127-141 Conv3x3).  Forward and the gradient w.r.t. x run the K10 / K17 / K11 / K13 kernels by shape (K10
    where >= 64 output channels fill the chip), the weight gradient K13 / K16 / K18 by channel counts (fixed-order sums); the
    remaining shapes are ATen/MIOpen."""
    if padding not in (0, 1, 2) or weight.shape[2:] != (3, 3):
        raise RuntimeError("conv3x3: 3x3 kernel with padding 0, 1 or 2 expected")
    if not x.is_cuda:
        raise RuntimeError("libdmh_hip ops need CUDA (ROCm) tensors; got device %s -- there is no CPU path" % x.device)
    return _Conv3x3.apply(_c(x), weight, bias, int(padding))


def _stem_wrw(x, g, weight, mean, std):
    """K21: dW of the 7x7/2 first convolution on (x - mean) / std, fixed-order sums on the fp32 MFMA; None when the shape is
    not the kernel's (3 -> 64 channels, even H, W a multiple of 8): the caller then takes ATen's."""
    B, Cin, H, W = x.shape
    if not WINO_ENABLED or not WRW_ENABLED or tuple(weight.shape) != (64, 3, 7, 7) or Cin != 3 or x.numel() * 4 > 0xFFFFFF00:
        return None
    lib = N.lib()
    n = lib.dmh_stem_wrw_workspace_size(B, H, W)
    if n < 0:
        return None
    ws = torch.empty(n, device=x.device, dtype=torch.float32)
    g_w = torch.empty_like(weight, memory_format=torch.contiguous_format)
    N.check(_timed("stem_wrw", lambda: lib.dmh_stem_wrw(N.ptr(x), N.ptr(g), B, H, W, mean, std, N.ptr(ws), N.ptr(g_w),
                                                        N.stream()), 4 * (x.numel() + g.numel()), 2 * 147 * g.numel()))
    return g_w


class _StemConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        ctx.params_const = _wino_frozen > 0
        return torch.conv2d(x, weight, None, 2, 3)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        lib = N.lib()
        g = _c(g)
        B, Cin, H, W = x.shape
        K = weight.shape[0]
        g_x = g_w = None
        need_w = ctx.needs_input_grad[1] and not ctx.params_const
        if ctx.needs_input_grad[0]:
            g_x = torch.empty_like(x)
            nb = 4 * (g.numel() + g_x.numel())
            N.check(_timed("stem_conv_bwd", lambda: lib.dmh_conv7x7s2_bwd_data(N.ptr(g), N.ptr(_c(weight.detach())), B, K,
                                                                              Cin, H, W, N.ptr(g_x), N.stream()), nb,
                           2 * 49 * Cin * g.numel()))
        if need_w:
            g_w = _stem_wrw(x, g, weight, 0.0, 1.0)
            if g_w is None:
                g_w = torch.ops.aten.convolution_backward(g, x, weight, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1,
                                                          [False, True, False])[1]
        return g_x, g_w


def stem_conv(x, weight):
    """nn.Conv2d(Cin, K, 7, stride=2, padding=3, bias=False) -- the encoder's first convolution
    (MD2/networks/resnet_encoder.py:88).  Forward is MIOpen, the weight gradient K21 (3 -> 64 channels; MIOpen otherwise); the gradient w.r.t. the image (what
    every attack step back-propagates to the patch) is the K12 gather kernel."""
    B, Cin, H, W = x.shape
    if (not x.is_cuda or not WINO_ENABLED or weight.shape[1:] != (Cin, 7, 7) or Cin > 4 or weight.shape[0] % 8 or H % 2
            or W % 2 or not x.requires_grad):
        return torch.conv2d(x, weight, None, 2, 3)
    return _StemConv.apply(_c(x), weight)


class _StemConvNorm(torch.autograd.Function):
    """K14 forward (normalisation + 7x7/2 convolution on the fp32 MFMA); backward: image gradient by K12 (scaled by 1/std),
    weight gradient by K21 (train pass only; ATen on the re-normalised image for widths that are not multiples of 8)."""

    @staticmethod
    def forward(ctx, x, weight, mean, std):
        lib = N.lib()
        B, _, H, W = x.shape
        y = torch.empty((B, 64, H // 2, W // 2), device=x.device, dtype=torch.float32)
        nb = 4 * (x.numel() + y.numel())
        N.check(_timed("stem_conv_fwd", lambda: lib.dmh_stem_conv_norm_fwd(N.ptr(x), N.ptr(_c(weight.detach())), B, H, W, mean,
                                                                          std, N.ptr(y), N.stream()), nb,
                       2 * 147 * y.numel()))
        ctx.save_for_backward(x, weight)
        ctx.norm = (mean, std)
        ctx.params_const = _wino_frozen > 0
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        mean, std = ctx.norm
        lib = N.lib()
        g = _c(g)
        B, Cin, H, W = x.shape
        g_x = g_w = None
        if ctx.needs_input_grad[0]:
            g_x = torch.empty_like(x)
            nb = 4 * (g.numel() + g_x.numel())
            # d/dx of conv((x - mean) / std, w) = conv_bwd_data(g, w / std): the 1/std goes into the 9,408 filter taps
            # (once per attack inside frozen_weights()) instead of a pass over the image gradient
            w_s = frozen_memo(("stem_w_over_std", weight.data_ptr(), weight._version, std),
                              lambda: _c(weight.detach() * (1.0 / std)))
            N.check(_timed("stem_conv_bwd", lambda: lib.dmh_conv7x7s2_bwd_data(N.ptr(g), N.ptr(w_s), B, 64, Cin, H, W,
                                                                              N.ptr(g_x), N.stream()), nb, 2 * 147 * g.numel()))
        if ctx.needs_input_grad[1] and not ctx.params_const:
            g_w = _stem_wrw(x, g, weight, mean, std)
            if g_w is None:
                xn = (x - mean) / std
                g_w = torch.ops.aten.convolution_backward(g, xn, weight, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1,
                                                          [False, True, False])[1]
        return g_x, g_w, None, None


def stem_conv_norm(x, weight, mean=0.45, std=0.225):
    """conv1((x - mean) / std) for the encoder's first layer -- MD2/networks/resnet_encoder.py:89-90 with torchvision's
    ResNet.conv1 (3 -> 64 channels, 7x7, stride 2, padding 3, no bias) -- as ONE launch of the K14 MFMA kernel: the
    normalisation is applied while the input tile is staged, so neither the normalised image nor a layout transpose
    ever exists in memory."""
    if not x.is_cuda:
        raise RuntimeError("libdmh_hip ops need CUDA (ROCm) tensors; got device %s -- there is no CPU path" % x.device)
    if tuple(weight.shape) != (64, 3, 7, 7) or x.dim() != 4 or x.shape[1] != 3 or x.shape[2] % 2 or x.shape[3] % 2:
        raise RuntimeError("stem_conv_norm: x must be [B,3,H,W] with even H, W and weight [64,3,7,7]")
    return _StemConvNorm.apply(_c(x), weight, float(mean), float(std))


DOWN_WIDE = os.environ.get("DMH_DOWN_WIDE", "1") != "0"     # A/B switch: K15 with 16-byte loaders and the filter image (round 5)


def _down_image(w3, wd):
    """The filter pair of K15 as an image of the kernels' LDS layout (csrc/down_conv.hip, "round 5"): [rows / 32][inner / 8][2560]
    floats from w3 [rows, inner, 3, 3] and wd [rows, inner(, 1, 1)] or None.  Once per weight tensor inside frozen_weights()."""
    lib = N.lib()
    rows, inner = w3.shape[0], w3.shape[1]

    def make():
        img = torch.empty(lib.dmh_down_conv_image_size(rows, inner), device=w3.device, dtype=torch.float32)
        N.check(lib.dmh_down_conv_weight_image(N.ptr(w3), N.ptr(wd), rows, inner, N.ptr(img), N.stream()))
        # the entry keeps its source tensors: a caller may hand in a temporary (a contiguous copy of a channels_last parameter),
        # and a freed temporary's address could come to name another layer's data with the same version inside one scope
        return img, w3, wd
    return frozen_memo(("down_img", w3.data_ptr(), w3._version, None if wd is None else (wd.data_ptr(), wd._version)), make)[0]


def _k15_fwd(lib, x, w3, wd, shift3, shiftd, relu3, B, Cin, Cout, H, W, y3, yd, stream):
    """dmh_down_conv_fwd_act, in its image form (16-byte loaders; bit-identical) where the rows are whole 16-byte words."""
    if DOWN_WIDE and W % 4 == 0 and Cout % 32 == 0:
        return lib.dmh_down_conv_fwd_img(N.ptr(x), N.ptr(_down_image(w3, wd)), int(wd is not None), N.ptr(shift3), N.ptr(shiftd),
                                         int(relu3), B, Cin, Cout, H, W, N.ptr(y3), N.ptr(yd), stream)
    return lib.dmh_down_conv_fwd_act(N.ptr(x), N.ptr(w3), N.ptr(wd), N.ptr(shift3), N.ptr(shiftd), int(relu3), B, Cin, Cout, H, W,
                                     N.ptr(y3), N.ptr(yd), stream)


def _k15_bwd(lib, g3, gd, w3t, wdt, g_add, B, Cin, Cout, H, W, g_x, stream):
    """dmh_down_conv_bwd_data_acc (filters transposed: w3t [Cin, Cout, 3, 3], wdt [Cin, Cout]), image form where W % 8 == 0."""
    if DOWN_WIDE and W % 8 == 0 and Cin % 32 == 0:
        return lib.dmh_down_conv_bwd_data_img(N.ptr(g3), N.ptr(gd), N.ptr(_down_image(w3t, wdt)), N.ptr(g_add), B, Cin, Cout, H, W,
                                              N.ptr(g_x), stream)
    return lib.dmh_down_conv_bwd_data_acc(N.ptr(g3), N.ptr(gd), N.ptr(w3t), N.ptr(wdt), N.ptr(g_add), B, Cin, Cout, H, W, N.ptr(g_x),
                                          stream)


class _DownConvs(torch.autograd.Function):
    """K15: conv3x3 stride 2 and the 1x1 stride-2 shortcut convolution of a down-sampling BasicBlock, one launch; backward:
    both input gradients in one launch (no separate accumulation pass), both weight gradients in one K20 launch (train pass
    only; ATen for input widths that are not multiples of 8)."""

    @staticmethod
    def forward(ctx, x, w3, wd):
        lib = N.lib()
        B, Cin, H, W = x.shape
        Cout = w3.shape[0]
        y3 = torch.empty((B, Cout, H // 2, W // 2), device=x.device, dtype=torch.float32)
        yd = torch.empty_like(y3)
        nb = 4 * (x.numel() + 2 * y3.numel() + w3.numel() + wd.numel())
        N.check(_timed("down_conv_fwd", lambda: _k15_fwd(lib, x, _c(w3.detach()), _c(wd.detach()), None, None, 0, B, Cin, Cout, H, W,
                                                         y3, yd, N.stream()), nb, 20 * Cin * y3.numel()))
        ctx.save_for_backward(x, w3, wd)
        ctx.params_const = _wino_frozen > 0
        return y3, yd

    @staticmethod
    def backward(ctx, g3, gd):
        x, w3, wd = ctx.saved_tensors
        lib = N.lib()
        B, Cin, H, W = x.shape
        Cout = w3.shape[0]
        g3, gd = _c(g3), _c(gd)
        g_x = g_w3 = g_wd = None
        if ctx.needs_input_grad[0]:
            # the kernel takes the filters transposed ([C_in][C_out][...]): once per attack inside frozen_weights()
            w3t = frozen_memo(("down_w3t", w3.data_ptr(), w3._version), lambda: _c(w3.detach().transpose(0, 1)))
            wdt = frozen_memo(("down_wdt", wd.data_ptr(), wd._version), lambda: _c(wd.detach().reshape(Cout, Cin).t()))
            g_x = torch.empty_like(x)
            nb = 4 * (g_x.numel() + 2 * g3.numel() + w3.numel() + wd.numel())
            N.check(_timed("down_conv_bwd", lambda: _k15_bwd(lib, g3, gd, w3t, wdt, None, B, Cin, Cout, H, W, g_x, N.stream()),
                           nb, 20 * Cin * g3.numel()))
        if not ctx.params_const and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            n = lib.dmh_down_wrw_workspace_size(B, Cin, Cout, H, W) if (WRW_ENABLED and x.numel() * 4 <= 0xFFFFFF00) else -1
            if n >= 0 and ctx.needs_input_grad[1]:
                ws = torch.empty(n, device=x.device, dtype=torch.float32)
                g_w3 = torch.empty_like(w3, memory_format=torch.contiguous_format)
                with_d = bool(ctx.needs_input_grad[2])
                g_wd = torch.empty_like(wd, memory_format=torch.contiguous_format) if with_d else None
                N.check(_timed("down_wrw", lambda: lib.dmh_down_wrw(
                    N.ptr(x), N.ptr(g3), N.ptr(gd) if with_d else None, B, Cin, Cout, H, W, N.ptr(ws), N.ptr(g_w3),
                    N.ptr(g_wd) if with_d else None, N.stream()), 4 * (x.numel() + 2 * g3.numel()), 20 * Cin * g3.numel()))
                return g_x, g_w3, g_wd
            if ctx.needs_input_grad[1]:
                g_w3 = torch.ops.aten.convolution_backward(g3, x, w3, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1,
                                                           [False, True, False])[1]
            if ctx.needs_input_grad[2]:
                g_wd = torch.ops.aten.convolution_backward(gd, x, wd, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1,
                                                           [False, True, False])[1]
        return g_x, g_w3, g_wd


def down_convs_ok(x, w3, wd):
    """Shapes K15 takes: fp32 CUDA, [C_out, C_in, 3, 3] + [C_out, C_in, 1, 1] with both channel counts multiples of 64,
    even input size."""
    return (WINO_ENABLED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and w3.dim() == 4 and wd.dim() == 4
            and tuple(w3.shape[2:]) == (3, 3) and tuple(wd.shape[2:]) == (1, 1) and w3.shape[:2] == wd.shape[:2]
            and w3.shape[1] == x.shape[1] and x.shape[1] % 64 == 0 and w3.shape[0] % 64 == 0
            and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and x.shape[2] >= 2 and x.shape[3] >= 2
            # the kernel's own 32-bit index limits (DMH_REQUIRE in csrc/down_conv.hip): beyond them the caller falls back
            and x.shape[0] * max(x.shape[1], w3.shape[0]) * x.shape[2] * x.shape[3] < (1 << 29)
            and x.shape[1] * w3.shape[0] < (1 << 24))


def down_convs(x, w3, wd):
    """(conv2d(x, w3, stride=2, padding=1), conv2d(x, wd, stride=2)): the first 3x3 convolution and the 1x1 shortcut of a
    down-sampling torchvision BasicBlock (layer2.0 / layer3.0 / layer4.0 under MD2/networks/resnet_encoder.py:94-98), which
    read the same input -- one K15 MFMA launch forward, one for the summed input gradient."""
    if not x.is_cuda:
        raise RuntimeError("libdmh_hip ops need CUDA (ROCm) tensors; got device %s -- there is no CPU path" % x.device)
    if not down_convs_ok(x, w3, wd):
        raise RuntimeError("down_convs: x [B,C,H,W] (C %% 64 == 0, even H, W), w3 [K,C,3,3], wd [K,C,1,1] (K %% 64 == 0) expected")
    return _DownConvs.apply(_c(x), w3, wd)


class _BasicBlockEval(torch.autograd.Function):
    """A stride-1 torchvision BasicBlock in eval() with constant parameters (inside an attack), as one autograd node:
        out1 = relu(bn1(conv1(x)));   y = relu(bn2(conv2(out1)) + x)          MD2/networks/resnet_encoder.py:85-98
    forward = the two K10 launches of conv3x3_bn_act; backward = one K9 pass (mask by y) + two K10 launches whose
    epilogues apply the ReLU mask of out1 and add the identity branch's gradient -- instead of two K9 passes, two K10
    launches and autograd's accumulation add."""

    @staticmethod
    def forward(ctx, x, w1, s1, b1, w2, s2, b2):
        lib = N.lib()
        B, Cc, H, W = x.shape
        out1, y = torch.empty_like(x), torch.empty_like(x)
        nb = 4 * 2 * x.numel()
        N.check(_timed("wino_conv3x3", lambda: _k10_act(lib, 
            N.ptr(x), N.ptr(_wino_filter(w1, False, s1)), N.ptr(b1), None, 1, B, Cc, Cc, H, W, 1, N.ptr(out1), N.stream()),
            nb, 18 * Cc * x.numel()))
        N.check(_timed("wino_conv3x3", lambda: _k10_act(lib, 
            N.ptr(out1), N.ptr(_wino_filter(w2, False, s2)), N.ptr(b2), N.ptr(x), 1, B, Cc, Cc, H, W, 1, N.ptr(y), N.stream()),
            nb + 4 * x.numel(), 18 * Cc * x.numel()))
        ctx.save_for_backward(out1, y, w1, s1, w2, s2)
        return y

    @staticmethod
    def backward(ctx, g):
        out1, y, w1, s1, w2, s2 = ctx.saved_tensors
        lib = N.lib()
        B, Cc, H, W = y.shape
        g = _c(g)
        ones = frozen_memo(("ones", Cc, g.device), lambda: torch.ones(Cc, device=g.device, dtype=torch.float32))
        g2 = torch.empty_like(g)            # g * [y > 0]: gradient of conv2's BatchNorm output and of the identity branch
        N.check(_timed("bn_act_bwd", lambda: lib.dmh_bn_act_bwd(N.ptr(y), N.ptr(g), N.ptr(ones), B, Cc, H * W, 1, N.ptr(g2),
                                                               None, N.stream()), 12 * g.numel()))
        g1, g_x = torch.empty_like(g), torch.empty_like(g)
        nb = 4 * 3 * g.numel()
        # conv2's backward-data, masked by [out1 > 0] in the epilogue (flag 2)
        N.check(_timed("wino_conv3x3", lambda: _k10_act(lib, 
            N.ptr(g2), N.ptr(_wino_filter(w2, True, s2)), None, N.ptr(out1), 2, B, Cc, Cc, H, W, 1, N.ptr(g1), N.stream()),
            nb, 18 * Cc * g.numel()))
        # conv1's backward-data + the identity branch's gradient in the epilogue
        N.check(_timed("wino_conv3x3", lambda: _k10_act(lib, 
            N.ptr(g1), N.ptr(_wino_filter(w1, True, s1)), None, N.ptr(g2), 0, B, Cc, Cc, H, W, 1, N.ptr(g_x), N.stream()),
            nb, 18 * Cc * g.numel()))
        return g_x, None, None, None, None, None, None


def basic_block_eval_ok(x, w1, w2):
    """Shapes and state the one-node BasicBlock takes: inside frozen_weights() (no parameter gradients exist there), fp32
    CUDA, equal channel counts, both convolutions on the K10 epilogue kernel (no channel split)."""
    if not (_wino_frozen > 0 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return False
    B, Cc, H, W = x.shape
    return (tuple(w1.shape) == (Cc, Cc, 3, 3) and tuple(w2.shape) == (Cc, Cc, 3, 3)
            and _wino_ok(B, Cc, Cc, H, W, allow_split=False, allow_sk=True))


def basic_block_eval(x, w1, scale1, shift1, w2, scale2, shift2):
    """relu(bn2(conv2(relu(bn1(conv1(x))))) + x) with the BatchNorms given as (scale, shift): a stride-1 BasicBlock of the
    encoder during an attack (model in eval(), parameters constant), as one autograd node (see _BasicBlockEval)."""
    if not basic_block_eval_ok(x, w1, w2):
        raise RuntimeError("basic_block_eval: needs ops.frozen_weights(), an fp32 CUDA [B,C,H,W] input and [C,C,3,3] filters "
                           "of a shape the Winograd kernel takes (ops.basic_block_eval_ok)")
    _reject_affine_grad("basic_block_eval", scale1, shift1)
    _reject_affine_grad("basic_block_eval", scale2, shift2)
    if not x.requires_grad:         # nothing to differentiate: the two fused launches
        y = conv3x3_bn_act(x, w1, scale1, shift1, None, True, 1)
        return conv3x3_bn_act(y, w2, scale2, shift2, x, True, 1)
    return _BasicBlockEval.apply(_c(x), w1.detach(), _c(scale1.detach()), _c(shift1.detach()), w2.detach(),
                                 _c(scale2.detach()), _c(shift2.detach()))


class _DownBlockEval(torch.autograd.Function):
    """A down-sampling torchvision BasicBlock in eval() with constant parameters (inside an attack), as one autograd node:
        out1 = relu(bn1(conv1_s2(x)));  idt = bn_d(conv1x1_s2(x));  y = relu(bn2(conv2(out1)) + idt)
    forward = ONE K15 launch (both BatchNorm scales folded into the filters, shifts and the ReLU in its epilogue) + one K10
    launch; backward = one K9 pass (mask by y), one K10 launch (conv2's backward-data masked by [out1 > 0] in the
    epilogue) and one K15 launch on the transposed, scaled filters -- no separate BatchNorm / ReLU passes."""

    @staticmethod
    def forward(ctx, x, w3, s1, b1, wd, sd, bd, w2, s2, b2, want_skip):
        lib = N.lib()
        B, Cin, H, W = x.shape
        Co = w3.shape[0]
        w3s = frozen_memo(("down_w3s", w3.data_ptr(), w3._version, s1.data_ptr()), lambda: _c(w3 * s1.view(-1, 1, 1, 1)))
        wds = frozen_memo(("down_wds", wd.data_ptr(), wd._version, sd.data_ptr()), lambda: _c(wd * sd.view(-1, 1, 1, 1)))
        out1 = torch.empty((B, Co, H // 2, W // 2), device=x.device, dtype=torch.float32)
        idt, y = torch.empty_like(out1), torch.empty_like(out1)
        N.check(_timed("down_conv_fwd", lambda: _k15_fwd(lib, x, w3s, wds, b1, bd, 1, B, Cin, Co, H, W, out1, idt, N.stream()),
                       4 * (x.numel() + 2 * out1.numel() + w3.numel() + wd.numel()), 20 * Cin * out1.numel()))
        N.check(_timed("wino_conv3x3", lambda: _k10_act(lib, 
            N.ptr(out1), N.ptr(_wino_filter(w2, False, s2)), N.ptr(b2), N.ptr(idt), 1, B, Co, Co, H // 2, W // 2, 1, N.ptr(y),
            N.stream()), 4 * 3 * out1.numel(), 18 * Co * out1.numel()))
        ctx.save_for_backward(out1, y, w3, s1, wd, sd, w2, s2)
        ctx.in_shape = (B, Cin, H, W)
        ctx.set_materialize_grads(False)
        # want_skip: x is handed back as a second output (an alias).  A caller that feeds THAT tensor to x's other consumer
        # (the decoder's skip connection) makes its gradient an input of this node's backward, where K15's epilogue adds it:
        # autograd's separate accumulation pass over the feature map disappears.
        return (y, x) if want_skip else (y, None)

    @staticmethod
    def backward(ctx, g, g_skip=None):
        out1, y, w3, s1, wd, sd, w2, s2 = ctx.saved_tensors
        lib = N.lib()
        B, Cin, H, W = ctx.in_shape
        Co = w3.shape[0]
        if g is None:           # only the alias was used
            return (g_skip,) + (None,) * 10
        g = _c(g)
        g_skip = None if g_skip is None else _c(g_skip)
        ones = frozen_memo(("ones", Co, g.device), lambda: torch.ones(Co, device=g.device, dtype=torch.float32))
        g2 = torch.empty_like(g)            # g * [y > 0]: gradient of bn2's output and of the shortcut's BatchNorm output
        N.check(_timed("bn_act_bwd", lambda: lib.dmh_bn_act_bwd(N.ptr(y), N.ptr(g), N.ptr(ones), B, Co, (H // 2) * (W // 2), 1,
                                                               N.ptr(g2), None, N.stream()), 12 * g.numel()))
        g1 = torch.empty_like(g)            # gradient of bn1's output: conv2's backward-data masked by [out1 > 0]
        N.check(_timed("wino_conv3x3", lambda: _k10_act(lib, 
            N.ptr(g2), N.ptr(_wino_filter(w2, True, s2)), None, N.ptr(out1), 2, B, Co, Co, H // 2, W // 2, 1, N.ptr(g1),
            N.stream()), 4 * 3 * g.numel(), 18 * Co * g.numel()))
        w3ts = frozen_memo(("down_w3ts", w3.data_ptr(), w3._version, s1.data_ptr()),
                           lambda: _c((w3 * s1.view(-1, 1, 1, 1)).transpose(0, 1)))
        wdts = frozen_memo(("down_wdts", wd.data_ptr(), wd._version, sd.data_ptr()),
                           lambda: _c((wd * sd.view(-1, 1, 1, 1)).reshape(Co, Cin).t()))
        g_x = torch.empty((B, Cin, H, W), device=g.device, dtype=torch.float32)
        N.check(_timed("down_conv_bwd", lambda: _k15_bwd(lib, g1, g2, w3ts, wdts, g_skip, B, Cin, Co, H, W, g_x, N.stream()),
                       4 * (g_x.numel() * (1 if g_skip is None else 2) + 2 * g.numel()), 20 * Cin * g.numel()))
        return (g_x,) + (None,) * 10


DOWN_NODE_ENABLED = os.environ.get("DMH_DOWN_NODE", "1") != "0"     # timing comparisons


def down_block_eval_ok(x, w3, wd, w2):
    """Inside frozen_weights(), K15 shapes, and the block's second convolution on the K10 epilogue kernel."""
    if not (_wino_frozen > 0 and DOWN_NODE_ENABLED and down_convs_ok(x, w3, wd)):
        return False
    Co = w3.shape[0]
    return tuple(w2.shape) == (Co, Co, 3, 3) and _wino_ok(x.shape[0], Co, Co, x.shape[2] // 2, x.shape[3] // 2, allow_split=False,
                                                         allow_sk=True)


def down_block_eval(x, w3, scale1, shift1, wd, scale_d, shift_d, w2, scale2, shift2, return_skip=False):
    """relu(bn2(conv2(relu(bn1(conv1_s2(x))))) + bn_d(conv1x1_s2(x))) with the BatchNorms given as (scale, shift): a
    down-sampling BasicBlock of the encoder during an attack, as one autograd node (see _DownBlockEval).
    ``return_skip``: returns (y, x') with x' an alias of x to be used by x's OTHER consumer (the decoder's skip connection,
    MD2/networks/depth_decoder.py:53-57): its gradient then enters this node and is added in K15's epilogue."""
    if not down_block_eval_ok(x, w3, wd, w2):
        raise RuntimeError("down_block_eval: needs ops.frozen_weights() and shapes K15 / K10 take (ops.down_block_eval_ok)")
    for sc, sh in ((scale1, shift1), (scale_d, shift_d), (scale2, shift2)):
        _reject_affine_grad("down_block_eval", sc, sh)
    d = lambda t: _c(t.detach())        # noqa: E731
    x = _c(x)
    args = (x, w3.detach(), d(scale1), d(shift1), wd.detach(), d(scale_d), d(shift_d), w2.detach(), d(scale2), d(shift2))
    if not x.requires_grad:
        with torch.no_grad():
            y, _ = _DownBlockEval.apply(*args, False)
        return (y, x) if return_skip else y
    y, skip = _DownBlockEval.apply(*args, bool(return_skip))
    return (y, skip) if return_skip else y


# ---------------------------------------------------------------------------------------------------------------
# K19: the attack's cost evaluated on one window per scene (roi.py plans the windows, csrc/roi_glue.hip is the pass
# between the convolutions).  The convolution kernels are the ones above, chosen here by what a kernel CAN take: a window
# launch is smaller than the shapes the throughput thresholds of _wino_ok / _wino32_ok were measured on.
# ---------------------------------------------------------------------------------------------------------------
ROI_ENABLED = os.environ.get("DMH_ROI", "1") != "0"      # A/B switch: 0 = the attack runs the whole frame


def _conv_any(x, weight, bias, pad, backward=False):
    """conv2d(x, weight, padding=pad) -- or, with ``backward``, the same filter's backward-data pass on a gradient x with
    pad' = 2 - pad_forward -- on whichever hand-written kernel takes the channel counts; ATen otherwise."""
    B, n_in, H, W = x.shape
    n_out = weight.shape[1] if backward else weight.shape[0]
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    if WINO_ENABLED and x.numel() < (1 << 30):
        if not backward and n_out == 1 and pad == 0 and n_in % 4 == 0:
            y = torch.empty((B, 1, Ho, Wo), device=x.device, dtype=torch.float32)
            N.check(_timed("conv3x3_head", lambda: N.lib().dmh_conv3x3_head(
                N.ptr(x), N.ptr(_c(weight.detach())), N.ptr(bias), B, n_in, H, W, 0, N.ptr(y), N.stream()),
                4 * (x.numel() + y.numel())))
            return y
        if backward and n_in == 1 and pad == 2 and n_out % 4 == 0 and n_out <= 64 and Ho >= 3 and Wo >= 3:
            y = torch.empty((B, n_out, Ho, Wo), device=x.device, dtype=torch.float32)
            N.check(_timed("conv3x3_head_bwd", lambda: N.lib().dmh_conv3x3_head_bwd_data(
                N.ptr(x), N.ptr(_c(weight.detach())), B, n_out, Ho, Wo, N.ptr(y), N.stream()), 4 * (x.numel() + y.numel())))
            return y
        if _small_ok(n_in, n_out):
            return _small_conv(x, weight, bias, pad, backward)
        if n_in % 8 == 0 and n_in >= 24 and Ho % 2 == 0 and Wo % 2 == 0 and Ho >= 2 and Wo >= 2:
            if n_out % 32 == 0 and n_out % 64 and n_out <= 96:
                return _wino32_conv(x, _wino32_filter(weight, backward), bias, n_out, pad)
            if n_out >= 64:
                return _wino_conv(x, _wino_filter(weight, backward), bias, n_out, pad)
    if backward:
        return torch.nn.functional.conv_transpose2d(x, weight, None, 1, 2 - pad)
    return torch.conv2d(x, weight, bias, 1, pad)


def _roi_glue_args(y, y_org, skip, skip_org, dst_org, size, frame, up, elu):
    a = N.RoiGlueArgs()
    a.y, a.skip = N.ptr(y), N.ptr(skip)
    a.y_org, a.skip_org, a.dst_org = N.ptr(y_org), N.ptr(skip_org), N.ptr(dst_org)
    a.B, a.C1, a.C2 = y.shape[0], y.shape[1], (0 if skip is None else skip.shape[1])
    a.sh, a.sw = y.shape[2], y.shape[3]
    a.kh, a.kw = (0, 0) if skip is None else (skip.shape[2], skip.shape[3])
    a.hc, a.wc = size
    a.H, a.W = frame
    a.up, a.elu = int(up), int(elu)
    return a


def _roi_glue_fwd(a, device):
    out = torch.empty((a.B, a.C1 + a.C2, a.hc + 2, a.wc + 2), device=device, dtype=torch.float32)
    N.check(_timed("roi_glue_fwd", lambda: N.lib().dmh_roi_glue_fwd(C.byref(a), N.ptr(out), N.stream()), 8 * out.numel()))
    return out


def _roi_glue_bwd(a, g_out, device, want_skip, y_region=None, skip_region=None, skip_prezero=True):
    """``y_region`` / ``skip_region`` = (org table [B,2], (rows, cols)): for a whole-frame source only that rectangle is
    computed; the rest of g_y is zero-filled, and so is the rest of g_skip unless ``skip_prezero`` is False (its consumer
    reads inside the rectangle only)."""
    # A rectangle needs the rest of its plane zero-filled first (one more launch, and every byte of the plane written once more):
    # where the plane is less than three times the rectangle the kernel takes the WHOLE plane instead and writes the zeros of
    # the elements no window entry reads itself (round 6: upconv(depth,0)'s output and encoder feature 3 at the attack's sizes)
    if y_region is not None and a.sh * a.sw < 3 * y_region[1][0] * y_region[1][1]:
        y_region = None
    if skip_region is not None and skip_prezero and a.kh * a.kw < 3 * skip_region[1][0] * skip_region[1][1]:
        skip_region = None
    new = torch.zeros if y_region is not None else torch.empty
    g_y = new((a.B, a.C1, a.sh, a.sw), device=device, dtype=torch.float32)
    g_skip = None
    if want_skip:
        new = torch.zeros if (skip_region is not None and skip_prezero) else torch.empty
        g_skip = new((a.B, a.C2, a.kh, a.kw), device=device, dtype=torch.float32)
    yo, (yh, yw) = y_region if y_region is not None else (None, (0, 0))
    ko, (kh, kw) = skip_region if (skip_region is not None and want_skip) else (None, (0, 0))
    nb = 4 * (g_out.numel() + 2 * a.B * a.C1 * (yh * yw if yo is not None else a.sh * a.sw) +
              (0 if g_skip is None else a.B * a.C2 * (kh * kw if ko is not None else a.kh * a.kw)))
    N.check(_timed("roi_glue_bwd", lambda: N.lib().dmh_roi_glue_bwd(C.byref(a), N.ptr(g_out), N.ptr(g_y), N.ptr(g_skip),
                                                                   N.ptr(yo), yh, yw, N.ptr(ko), kh, kw, N.stream()), nb))
    return g_y, g_skip


from .roi import STAGES as _ROI_STAGES, TABLE as _ROI_NAMES     # noqa: E402  (row order of the device table of origins)

ROI_DEPTH = int(os.environ.get("DMH_ROI_DEPTH", "4"))      # level of the deepest windowed decoder stage (2, 3 or 4)
# channel plan of the reference decoder's stages (MD2/networks/depth_decoder.py:22-37), by window name: (out, in)
_ROI_CHANNELS = {"d": (1, 16), "z01": (16, 16), "y00": (16, 32), "z11": (32, 96), "y10": (32, 64), "z21": (64, 128),
                 "y20": (64, 128), "z31": (128, 256), "y30": (128, 256), "z41": (256, 512)}


def _roi_chain(depth):
    """The windowed stages from z{depth}1 down to the disparity head, in execution order: (name, level, up, skip index)."""
    names = [st[0] for st in _ROI_STAGES]
    return list(reversed(_ROI_STAGES[:names.index("z%d1" % depth) + 1]))


class _RoiTail(torch.autograd.Function):
    """mean((disp0 * mask)^2) from upconv(depth,0)'s whole-frame output and the encoder features below it, through
    upconv(depth,1) ... dispconv(0) (MD2/networks/depth_decoder.py:51-63) evaluated on one window per scene.  Hand-written
    backward (parameters are constants: inside ops.frozen_weights() only): gradients w.r.t. that output and the features,
    zero outside the rectangles the windows reach.  Tensor arguments: x_top, feat_0 .. feat_{depth-1}, mask, tab, then
    (weight, bias) per stage in execution order."""

    @staticmethod
    def _glue(chain, k, src, feats, org, sz, H0, W0, f0_compact):
        name, lvl, up, skip = chain[k]
        src_org = None if k == 0 else org[chain[k - 1][0]]
        skip_org = org["hz"] if (skip == 0 and f0_compact) else None      # feature 0 handed on as its "hz" window only
        return _roi_glue_args(src, src_org, None if skip is None else feats[skip], skip_org, org[name], sz[name],
                              (H0 >> lvl, W0 >> lvl), up, 1)

    @staticmethod
    def forward(ctx, plan, depth, x_top, *rest):
        lib = N.lib()
        dev = x_top.device
        depth, sign = (depth if isinstance(depth, tuple) else (depth, 1.0))       # (depth, -1.0): the negated cost
        feats, mask, tab = rest[:depth], rest[depth], rest[depth + 1]
        wb = rest[depth + 2:]
        chain = _roi_chain(depth)
        B = x_top.shape[0]
        H0, W0 = x_top.shape[2] << (depth + 1), x_top.shape[3] << (depth + 1)
        f0c = bool(plan.f0_compact)
        if (any(tuple(f.shape[2:]) != ((H0 >> (k + 1), W0 >> (k + 1)) if (k or not f0c) else plan.size["hz"])
                for k, f in enumerate(feats))
                or tuple(mask.shape) != (B, 1, H0, W0) or plan.B != B or (plan.H, plan.W) != (H0, W0)):
            raise RuntimeError("roi tail: feature / mask / plan shapes do not match")
        org = {n: tab[k] for k, n in enumerate(_ROI_NAMES)}
        sz = plan.size
        outs, src = [], x_top
        for k in range(len(chain)):
            a = _RoiTail._glue(chain, k, src, feats, org, sz, H0, W0, f0c)
            src = _conv_any(_roi_glue_fwd(a, dev), wb[2 * k], wb[2 * k + 1], 0)
            outs.append(src)
        d_pre = outs.pop()
        hd, wd_ = sz["d"]
        sig = torch.empty_like(d_pre)
        part = torch.empty(lib.dmh_roi_cost_partials_size(B, hd, wd_), device=dev, dtype=torch.float32)
        cost = torch.empty((), device=dev, dtype=torch.float32)
        N.check(lib.dmh_roi_cost_fwd_scaled(N.ptr(d_pre), N.ptr(mask), N.ptr(org["d"]), B, hd, wd_, H0, W0, sign, N.ptr(sig),
                                            N.ptr(part), N.ptr(cost), N.stream()))
        ctx.save_for_backward(x_top, mask, tab, sig, *feats, *outs, *wb[0::2])
        ctx.plan, ctx.depth, ctx.f0c, ctx.sign = plan, depth, f0c, sign
        return cost

    @staticmethod
    def backward(ctx, g):
        plan, depth = ctx.plan, ctx.depth
        sv = ctx.saved_tensors
        x_top, mask, tab, sig = sv[:4]
        chain = _roi_chain(depth)
        n = len(chain)
        feats, outs, ws = sv[4:4 + depth], sv[4 + depth:4 + depth + n - 1], sv[4 + depth + n - 1:]
        lib = N.lib()
        dev = x_top.device
        B = x_top.shape[0]
        H0, W0 = plan.H, plan.W
        org = {nm: tab[k] for k, nm in enumerate(_ROI_NAMES)}
        sz = plan.size
        hd, wd_ = sz["d"]
        g_cur = torch.empty_like(sig)
        N.check(lib.dmh_roi_cost_bwd_scaled(N.ptr(sig), N.ptr(mask), N.ptr(org["d"]), B, hd, wd_, H0, W0, ctx.sign,
                                            N.ptr(_c(g.to(torch.float32))), N.ptr(g_cur), N.stream()))
        g_feats = [None] * depth
        for k in range(n - 1, -1, -1):
            name, lvl, up, skip = chain[k]
            src = x_top if k == 0 else outs[k - 1]
            a = _RoiTail._glue(chain, k, src, feats, org, sz, H0, W0, ctx.f0c)
            want_skip = skip is not None and ctx.needs_input_grad[3 + skip]
            y_reg = ("r_y%d0" % lvl) if k == 0 else None
            s_reg = None if (skip is None or (skip == 0 and ctx.f0c)) else "r_f%d" % skip      # a compact feature: all of it
            # feature 0's gradient is zero outside "r_f0"; an encoder head that runs its backward on the plan's windows reads
            # it inside that rectangle only, so the rest of the 252 MB tensor is not even zero-filled then
            g_cur, g_skip = _roi_glue_bwd(a, _conv_any(g_cur, ws[k], None, 2, True), dev, want_skip,
                                          y_region=None if y_reg is None else (org[y_reg], sz[y_reg]),
                                          skip_region=None if s_reg is None else (org[s_reg], sz[s_reg]),
                                          skip_prezero=not (skip == 0 and plan.head_windowed))
            if skip is not None:
                g_feats[skip] = g_skip
        return (None, None, g_cur) + tuple(g_feats) + (None,) * (2 + 2 * n)


def roi_tail_ok(x_top, feats, convs, depth):
    """The cropped tail applies: inside frozen_weights() (its backward has no parameter gradients), fp32 CUDA tensors, the
    reference decoder's channel plan for the stages upconv(depth,1) ... dispconv(0)."""
    if depth not in (2, 3, 4) or len(feats) != depth:
        return False
    chain = _roi_chain(depth)
    shapes = [tuple(c.weight.shape) for c in convs]
    return (ROI_ENABLED and _wino_frozen > 0 and x_top.is_cuda and all(t.dtype == torch.float32 for t in (x_top,) + tuple(feats))
            and shapes == [_ROI_CHANNELS[st[0]] + (3, 3) for st in chain]
            and x_top.shape[1] == _ROI_CHANNELS["z%d1" % depth][0]
            and all(f.shape[1] == _ROI_CHANNELS["z%d1" % (k + 1)][1] - _ROI_CHANNELS["z%d1" % (k + 1)][0] for k, f in enumerate(feats))
            and x_top.shape[2] >= 2 and x_top.shape[3] >= 2)


def roi_tail_cost(x_top, feats, mask, plan, tab, convs, negate=False):
    """mean((sigmoid(dispconv0(...)) * mask)^2) of the decoder tail on the windows of ``plan`` (roi.RoiPlan; ``tab`` is
    its origin table on the device, int32 [len(roi.TABLE), B, 2]).  ``x_top``: upconv(depth,0)'s whole-frame output (before
    its ELU), depth = plan.depth; ``feats``: encoder features 0 .. depth-1; ``convs``: the nn.Conv2d modules of upconv(depth,1)
    ... upconv(0,1), dispconv(0) in execution order.  Equals ops.masked_sq_mean(decoder(...)[("disp", 0)], mask) when the
    mask is zero outside the plan's boxes; ``negate``: minus that (what an attack hands to autograd), the sign applied inside
    the cost kernels."""
    depth = plan.depth
    feats = tuple(feats)
    if not roi_tail_ok(x_top, feats, convs, depth):
        raise RuntimeError("roi_tail_cost: needs ops.frozen_weights(), fp32 CUDA tensors and the Monodepth2 decoder tail")
    if tab.dtype != torch.int32 or tuple(tab.shape) != (len(_ROI_NAMES), x_top.shape[0], 2):
        raise RuntimeError("roi_tail_cost: origin table must be int32 [%d, B, 2] (roi.TABLE)" % len(_ROI_NAMES))
    wb = []
    for c in convs:
        wb += [c.weight.detach(), None if c.bias is None else _c(c.bias.detach())]
    return _RoiTail.apply(plan, (depth, -1.0) if negate else depth, _c(x_top), *[_c(f) for f in feats], _c(mask), _c(tab), *wb)


def _roi_crop(src, gate, g, org, size):
    """Compact [B, C, h, w] window of a whole-frame tensor (roi_crop / roi_mask of csrc/roi_encoder.hip)."""
    ref = src if src is not None else gate
    B, Cc, H, W = ref.shape
    out = torch.empty((B, Cc) + tuple(size), device=ref.device, dtype=torch.float32)
    N.check(_timed("roi_crop", lambda: N.lib().dmh_roi_crop(N.ptr(src), N.ptr(gate), N.ptr(g), N.ptr(org), B, Cc, H, W, size[0],
                                                           size[1], 0, N.ptr(out), N.stream()),
                   4 * out.numel() * (2 + (gate is not None))))
    return out


class _EncHeadEval(torch.autograd.Function):
    """conv1((x - 0.45) / 0.225) -> bn1 -> relu -> maxpool -> layer1 (two stride-1 BasicBlocks) of the ResNet encoder in
    eval() with constant parameters (MD2/networks/resnet_encoder.py:85-98), as ONE autograd node whose backward runs on the
    windows of a roi.RoiPlan: the attack reads d cost / d image under the pasted object only.  Forward = the same launches as
    the separate nodes (K14, K9 stem, four K10 launches).  Backward: layer1 on the "l1" window of the 1/4 map (four K10
    launches on compact tensors; each spoils one ring of the window, which is four rings larger than what is read of it),
    the stem's max-pool / ReLU / BatchNorm adjoint on the "gz" window of the 1/2 map, K12 on the image window "d"."""

    @staticmethod
    def forward(ctx, x, plan, tab, w_stem, s0, b0, w1a, s1a, b1a, w2a, s2a, b2a, w1b, s1b, b1b, w2b, s2b, b2b):
        lib = N.lib()
        B, _, H, W = x.shape
        dev = x.device
        st = N.stream()
        z = torch.empty((B, 64, H // 2, W // 2), device=dev, dtype=torch.float32)
        N.check(_timed("stem_conv_fwd", lambda: lib.dmh_stem_conv_norm_fwd(N.ptr(x), N.ptr(_c(w_stem)), B, H, W, 0.45, 0.225,
                                                                          N.ptr(z), st), 4 * (x.numel() + z.numel()),
                       2 * 147 * z.numel()))
        f0 = torch.empty_like(z)
        pooled = torch.empty((B, 64, H // 4, W // 4), device=dev, dtype=torch.float32)
        arg = torch.empty((B, 64, H // 4, W // 4), device=dev, dtype=torch.uint8)
        N.check(_timed("stem_fwd", lambda: lib.dmh_stem_bn_relu_pool_fwd(N.ptr(z), N.ptr(s0), N.ptr(b0), B, 64, H // 2, W // 2,
                                                                        N.ptr(f0), N.ptr(pooled), N.ptr(arg), st),
                       4 * (2 * z.numel() + pooled.numel()) + arg.numel()))
        del z

        def conv_act(inp, w, s, b, res):
            y = torch.empty_like(inp)
            N.check(_timed("wino_conv3x3", lambda: lib.dmh_wino_conv3x3_act(
                N.ptr(inp), N.ptr(_wino_filter(w, False, s)), N.ptr(b), N.ptr(res), 1, B, 64, 64, H // 4, W // 4, 1, N.ptr(y), st),
                4 * (2 + (res is not None)) * inp.numel(), 18 * 64 * inp.numel()))
            return y

        o1a = conv_act(pooled, w1a, s1a, b1a, None)
        ya = conv_act(o1a, w2a, s2a, b2a, pooled)
        o1b = conv_act(ya, w1b, s1b, b1b, None)
        f1 = conv_act(o1b, w2b, s2b, b2b, ya)
        ctx.save_for_backward(f0, arg, s0, o1a, ya, o1b, f1, tab, w_stem, w1a, s1a, w2a, s2a, w1b, s1b, w2b, s2b)
        ctx.plan, ctx.img = plan, (H, W)
        ctx.set_materialize_grads(False)
        return f0, f1

    @staticmethod
    def backward(ctx, g_f0, g_f1):
        f0, arg, s0, o1a, ya, o1b, f1, tab, w_stem, w1a, s1a, w2a, s2a, w1b, s1b, w2b, s2b = ctx.saved_tensors
        plan, (H, W) = ctx.plan, ctx.img
        lib = N.lib()
        dev = f0.device
        B = f0.shape[0]
        st = N.stream()
        org = {n: tab[k] for k, n in enumerate(_ROI_NAMES)}
        sz = plan.size
        hq, wq = sz["l1"]
        if g_f1 is None:
            g_pool = torch.zeros((B, 64, hq, wq), device=dev, dtype=torch.float32)
        else:
            def conv_bwd(g, w, s, res, flag):
                y = torch.empty_like(g)
                N.check(_timed("wino_conv3x3", lambda: lib.dmh_wino_conv3x3_act(
                    N.ptr(g), N.ptr(_wino_filter(w, True, s)), None, N.ptr(res), flag, B, 64, 64, hq, wq, 1, N.ptr(y), st),
                    4 * 3 * g.numel(), 18 * 64 * g.numel()))
                return y

            g2b = _roi_crop(_c(g_f1), f1, None, org["l1"], (hq, wq))                    # g * [y_b > 0] on the window
            g1b = conv_bwd(g2b, w2b, s2b, _roi_crop(o1b, None, None, org["l1"], (hq, wq)), 2)      # ... * [out1_b > 0]
            g_ya = conv_bwd(g1b, w1b, s1b, g2b, 0)                                      # + the identity branch
            g2a = _roi_crop(None, ya, g_ya, org["l1"], (hq, wq))                        # * [y_a > 0]
            g1a = conv_bwd(g2a, w2a, s2a, _roi_crop(o1a, None, None, org["l1"], (hq, wq)), 2)
            g_pool = conv_bwd(g1a, w1a, s1a, g2a, 0)
        hs, ws = sz["gz"]
        g_z = torch.empty((B, 64, hs, ws), device=dev, dtype=torch.float32)
        N.check(_timed("stem_bwd_win", lambda: lib.dmh_stem_bn_relu_pool_bwd_win(
            N.ptr(f0), N.ptr(arg), N.ptr(None if g_f0 is None else _c(g_f0)), N.ptr(g_pool), N.ptr(s0), N.ptr(org["gz"]),
            N.ptr(org["l1"]), B, 64, H // 2, W // 2, hs, ws, hq, wq, N.ptr(g_z), st), 4 * 4 * g_z.numel()))
        w_s = frozen_memo(("stem_w_over_std", w_stem.data_ptr(), w_stem._version, 0.225),
                          lambda: _c(w_stem.detach() * (1.0 / 0.225)))
        g_x = torch.zeros((B, 3, H, W), device=dev, dtype=torch.float32)
        hd, wd = sz["d"]
        N.check(_timed("stem_conv_bwd_win", lambda: lib.dmh_conv7x7s2_bwd_data_win(
            N.ptr(g_z), N.ptr(w_s), N.ptr(org["d"]), N.ptr(org["gz"]), B, 64, 3, H, W, hd, wd, hs, ws, N.ptr(g_x), st),
            4 * (g_z.numel() + B * 3 * hd * wd), 2 * 147 * 64 * B * (hd // 2) * (wd // 2)))
        return (g_x,) + (None,) * 17


def encoder_head_ok(x, conv1_weight, blocks):
    """The windowed encoder head applies: inside frozen_weights(), an fp32 CUDA image [B,3,H,W] with H, W multiples of 8, the
    standard 3 -> 64 7x7 first layer and a layer1 of two stride-1 64-channel BasicBlocks whose convolutions K10 takes."""
    if not (ROI_ENABLED and _wino_frozen > 0 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3
            and x.shape[2] % 8 == 0 and x.shape[3] % 8 == 0 and x.requires_grad and tuple(conv1_weight.shape) == (64, 3, 7, 7)):
        return False
    B, _, H, W = x.shape
    return (len(blocks) == 2 and all(tuple(w.shape) == (64, 64, 3, 3) for blk in blocks for w in blk)
            and _wino_ok(B, 64, 64, H // 4, W // 4, allow_split=False) and B * 64 * (H // 4) * (W // 4) < (1 << 30))


def encoder_head_eval(x, plan, tab, conv1_weight, aff0, blocks):
    """(feature 0, feature 1) of the ResNet-18 encoder in eval() -- MD2/networks/resnet_encoder.py:88-95: relu(bn1(conv1((x -
    0.45) / 0.225))) and layer1(maxpool(.)) -- as one node whose backward runs on the windows of ``plan`` (see _EncHeadEval):
    d / d x is exact inside the plan's image window "d" and ZERO outside it, which is all an object attack reads
    (physicalTrans.py:156-165).  ``blocks``: per BasicBlock (w1, (scale1, shift1), w2, (scale2, shift2)); ``aff0``: bn1's
    (scale, shift).  Marks the plan as head_windowed (the decoder tail then writes feature 0's gradient inside "r_f0" only)."""
    ws = [(b[0], b[2]) for b in blocks]
    if not encoder_head_ok(x, conv1_weight, ws):
        raise RuntimeError("encoder_head_eval: needs ops.frozen_weights() and the ResNet-18 head (ops.encoder_head_ok)")
    if tab.dtype != torch.int32 or tuple(tab.shape) != (len(_ROI_NAMES), x.shape[0], 2) or plan.B != x.shape[0] or \
            (plan.H, plan.W) != tuple(x.shape[2:]):
        raise RuntimeError("encoder_head_eval: plan / origin table do not match the image batch")
    d = lambda t: _c(t.detach())        # noqa: E731
    args = [d(conv1_weight), d(aff0[0]), d(aff0[1])]
    for w1, a1, w2, a2 in blocks:
        args += [w1.detach(), d(a1[0]), d(a1[1]), w2.detach(), d(a2[0]), d(a2[1])]
    plan.head_windowed = True
    return _EncHeadEval.apply(_c(x), plan, _c(tab), *args)


class CleanHead(object):
    """What the incremental encoder head keeps of the CLEAN scenes for the length of one attack: encoder feature 1 (layer1's
    output) of the un-pasted frames, twice -- ``pristine`` is never written; ``work`` is the tensor handed to the rest of the
    network, into which each attack step writes the cells the pasted object changes ("f1s") and puts the clean values back
    before the next step."""

    def __init__(self, f1, f2=None, frames=None):
        self.frames = frames        # the clean frames themselves: kept alive, so that their address cannot name other data
        self.generation = 0         # bumped by every step that writes into `work`: a backward of an older step must not run
        self.pristine, self.work = f1, f1.clone()
        self.dirty = None           # (origin table [B,2], (rows, cols)) of the window written by the last step
        self.pristine2, self.work2 = f2, (None if f2 is None else f2.clone())     # the same for feature 2 (layer2's output)
        self.dirty2 = None
        self._own = {}              # private copies of the origins of `dirty` / `dirty2` (mark(..., keep=True))

    def mark(self, which, org, size, keep):
        """Record the window step ``which`` (1: feature 1, 2: feature 2) has just written.  ``keep``: the origin table will be
        OVERWRITTEN before restore() reads it (an attack replayed from a HIP graph keeps one table buffer and copies every
        step's origins into it: torchattacks/attacks/phy_obj_atk.py, _graph_steps) -- the origins are then copied into a
        buffer of this object, by a device copy that is part of the step."""
        if keep:
            own = self._own.get(which)
            if own is None or own.shape != org.shape:
                own = self._own[which] = torch.empty_like(org)
            own.copy_(org)
            org = own
        if which == 1:
            self.dirty = (org, size)
        else:
            self.dirty2 = (org, size)

    @staticmethod
    def _put_back(pristine, work, dirty):
        org, (h, w) = dirty
        B, Cc, H, W = work.shape
        N.check(N.lib().dmh_roi_paste(N.ptr(pristine), None, H, W, N.ptr(org), B, Cc, H, W, h, w, N.ptr(work), N.stream()))

    def restore(self):
        if self.dirty is not None:
            self._put_back(self.pristine, self.work, self.dirty)
            self.dirty = None
        if self.dirty2 is not None:
            self._put_back(self.pristine2, self.work2, self.dirty2)
            self.dirty2 = None


class _EncHeadInc(torch.autograd.Function):
    """The encoder head of an attack step, incrementally: the pasted scene differs from the clean scene inside the object's
    box only (physicalTrans.py:156-165), so conv1 -> bn1 -> relu -> maxpool -> layer1 (MD2/networks/resnet_encoder.py:85-98)
    are evaluated on ONE compact window per scene ("hz" on the 1/2 map, "hl" on the 1/4 map: K14's window form, then the
    whole-tensor kernels on the compact tensors; zero padding / pooling padding at the window's edge spoil one ring per stage
    and the plan makes the window that much larger than what is read of it).  Outputs: feature 0 as its "hz" window (its only
    consumers, the decoder tail and this node's backward, read inside it) and feature 1 as the cached clean feature with the
    changed cells ("f1s") written in.  Backward: the same compact tensors through layer1's four backward-data launches, the
    stem's adjoint and K12's window form -- d / d image inside the plan's image window "d", zero elsewhere."""

    @staticmethod
    def forward(ctx, x, plan, tab, clean, w_stem, s0, b0, w1a, s1a, b1a, w2a, s2a, b2a, w1b, s1b, b1b, w2b, s2b, b2b, *l2):
        """``l2`` (optional, 15 tensors): layer2 = a down-sampling block (w3, s1, b1, wd, sd, bd, w2, s2, b2) and a stride-1
        block (w1, s1, b1, w2, s2, b2) -- evaluated on the plan's "h3" window and returned as a third output."""
        lib = N.lib()
        B, _, H, W = x.shape
        dev = x.device
        st = N.stream()
        org = {n: tab[k] for k, n in enumerate(_ROI_NAMES)}
        hl, wl = plan.size["hl"]
        hz, wz = 2 * hl, 2 * wl
        z = torch.empty((B, 64, hz, wz), device=dev, dtype=torch.float32)
        N.check(_timed("stem_conv_fwd", lambda: lib.dmh_stem_conv_norm_fwd_win(
            N.ptr(x), N.ptr(_c(w_stem)), N.ptr(org["hz"]), B, H, W, hz, wz, 0.45, 0.225, N.ptr(z), st),
            4 * (z.numel() + 12 * B * hz * wz), 2 * 147 * z.numel()))
        f0 = torch.empty_like(z)
        pooled = torch.empty((B, 64, hl, wl), device=dev, dtype=torch.float32)
        arg = torch.empty((B, 64, hl, wl), device=dev, dtype=torch.uint8)
        N.check(_timed("stem_fwd", lambda: lib.dmh_stem_bn_relu_pool_fwd(N.ptr(z), N.ptr(s0), N.ptr(b0), B, 64, hz, wz, N.ptr(f0),
                                                                        N.ptr(pooled), N.ptr(arg), st),
                       4 * (2 * z.numel() + pooled.numel()) + arg.numel()))
        del z

        def conv_act(inp, w, s, b, res):
            y = torch.empty_like(inp)
            N.check(_timed("wino_conv3x3", lambda: lib.dmh_wino_conv3x3_act(
                N.ptr(inp), N.ptr(_wino_filter(w, False, s)), N.ptr(b), N.ptr(res), 1, B, 64, 64, hl, wl, 1, N.ptr(y), st),
                4 * (2 + (res is not None)) * inp.numel(), 18 * 64 * inp.numel()))
            return y

        o1a = conv_act(pooled, w1a, s1a, b1a, None)
        ya = conv_act(o1a, w2a, s2a, b2a, pooled)
        o1b = conv_act(ya, w1b, s1b, b1b, None)
        f1c = conv_act(o1b, w2b, s2b, b2b, ya)
        # feature 1 = the clean scenes' feature with the changed cells written in (the clean values return before the next step)
        clean.restore()
        clean.generation += 1
        ctx.clean, ctx.generation = clean, clean.generation
        hs_, ws_ = plan.size["f1s"]
        N.check(_timed("roi_paste", lambda: lib.dmh_roi_paste(N.ptr(f1c), N.ptr(org["hl"]), hl, wl, N.ptr(org["f1s"]), B, 64,
                                                             H // 4, W // 4, hs_, ws_, N.ptr(clean.work), st),
                       8 * B * 64 * hs_ * ws_))
        clean.mark(1, org["f1s"], (hs_, ws_), getattr(plan, "table_rewritten", False))
        saved = [f0, arg, s0, o1a, ya, o1b, f1c, tab, w_stem, w1a, s1a, w2a, s2a, w1b, s1b, w2b, s2b]
        ctx.plan, ctx.img, ctx.l2 = plan, (H, W), bool(l2)
        ctx.set_materialize_grads(False)
        if not l2:
            ctx.save_for_backward(*saved)
            return f0, clean.work.detach()  # a fresh alias per step: the cached tensor itself never enters an autograd graph
        # ---- layer2 on the "h3" window of the 1/8 map: its input is the "h3in" window of feature 1 as just assembled
        w3, sc1, sh1, wd, scd, shd, w2, sc2, sh2, v1, t1, u1, v2, t2, u2 = l2
        h3, w3_ = plan.size["h3"]
        x2 = torch.empty((B, 64, 2 * h3, 2 * w3_), device=dev, dtype=torch.float32)
        N.check(_timed("roi_crop", lambda: lib.dmh_roi_crop(N.ptr(clean.work), None, None, N.ptr(org["h3in"]), B, 64, H // 4,
                                                           W // 4, 2 * h3, 2 * w3_, 0, N.ptr(x2), st), 8 * x2.numel()))
        w3s = frozen_memo(("down_w3s", w3.data_ptr(), w3._version, sc1.data_ptr()), lambda: _c(w3 * sc1.view(-1, 1, 1, 1)))
        wds = frozen_memo(("down_wds", wd.data_ptr(), wd._version, scd.data_ptr()), lambda: _c(wd * scd.view(-1, 1, 1, 1)))
        Co = w3.shape[0]
        q1 = torch.empty((B, Co, h3, w3_), device=dev, dtype=torch.float32)
        idt = torch.empty_like(q1)
        N.check(_timed("down_conv_fwd", lambda: _k15_fwd(lib, x2, w3s, wds, sh1, shd, 1, B, 64, Co, 2 * h3, 2 * w3_, q1, idt, st),
                       4 * (x2.numel() + 2 * q1.numel()), 20 * 64 * q1.numel()))

        def conv_act2(inp, w, sc, sh, res):
            y = torch.empty_like(inp)
            N.check(_timed("wino_conv3x3", lambda: lib.dmh_wino_conv3x3_act(
                N.ptr(inp), N.ptr(_wino_filter(w, False, sc)), N.ptr(sh), N.ptr(res), 1, B, Co, Co, h3, w3_, 1, N.ptr(y), st),
                4 * (2 + (res is not None)) * inp.numel(), 18 * Co * inp.numel()))
            return y

        y2a = conv_act2(q1, w2, sc2, sh2, idt)
        q2 = conv_act2(y2a, v1, t1, u1, None)
        f2c = conv_act2(q2, v2, t2, u2, y2a)
        hf, wf = plan.size["f2s"]
        N.check(_timed("roi_paste", lambda: lib.dmh_roi_paste(N.ptr(f2c), N.ptr(org["h3"]), h3, w3_, N.ptr(org["f2s"]), B, Co,
                                                             H // 8, W // 8, hf, wf, N.ptr(clean.work2), st), 8 * B * Co * hf * wf))
        clean.mark(2, org["f2s"], (hf, wf), getattr(plan, "table_rewritten", False))
        ctx.save_for_backward(*saved, q1, y2a, q2, f2c, w3, sc1, wd, scd, w2, sc2, v1, t1, v2, t2)
        return f0, clean.work.detach(), clean.work2.detach()

    @staticmethod
    def backward(ctx, g_f0, g_f1, g_f2=None):
        if ctx.clean.generation != ctx.generation:
            raise RuntimeError("encoder_head_incremental: a later forward has re-written the cached clean features this graph's "
                               "feature 1 / 2 alias; run each step's backward before the next step's forward")
        sv = ctx.saved_tensors
        f0, arg, s0, o1a, ya, o1b, f1c, tab, w_stem, w1a, s1a, w2a, s2a, w1b, s1b, w2b, s2b = sv[:17]
        plan, (H, W) = ctx.plan, ctx.img
        lib = N.lib()
        dev = f0.device
        B = f0.shape[0]
        st = N.stream()
        org = {n: tab[k] for k, n in enumerate(_ROI_NAMES)}
        hl, wl = plan.size["hl"]
        hz, wz = 2 * hl, 2 * wl
        g_pool = None
        f1_frame, f1_org = (H // 4, W // 4), org["hl"]          # where feature 1's gradient lives: the whole 1/4 map ...
        if ctx.l2 and g_f2 is not None:
            # ---- layer2's backward on "h3": feature 1's gradient comes out as the "h3in" window (which holds "hl")
            q1, y2a, q2, f2c, w3, sc1, wd, scd, w2, sc2, v1, t1, v2, t2 = sv[17:]
            h3, w3_ = plan.size["h3"]
            Co = w3.shape[0]

            def conv_bwd2(g, w, sc, res, flag):
                y = torch.empty_like(g)
                N.check(_timed("wino_conv3x3", lambda: lib.dmh_wino_conv3x3_act(
                    N.ptr(g), N.ptr(_wino_filter(w, True, sc)), None, N.ptr(res), flag, B, Co, Co, h3, w3_, 1, N.ptr(y), st),
                    4 * 3 * g.numel(), 18 * Co * g.numel()))
                return y

            gb = torch.empty_like(f2c)                                     # g * [f2 > 0] on the window
            N.check(_timed("roi_crop", lambda: lib.dmh_roi_crop(N.ptr(_c(g_f2)), N.ptr(f2c), None, N.ptr(org["h3"]), B, Co, H // 8,
                                                               W // 8, h3, w3_, 1, N.ptr(gb), st), 12 * gb.numel()))
            g1b = conv_bwd2(gb, v2, t2, q2, 2)
            g_y2a = conv_bwd2(g1b, v1, t1, gb, 0)
            g2 = torch.empty_like(g_y2a)                                   # * [y2a > 0]: bn2's output and the shortcut's
            N.check(_timed("roi_crop", lambda: lib.dmh_roi_crop(None, N.ptr(y2a), N.ptr(g_y2a), N.ptr(org["h3"]), B, Co, H // 8,
                                                               W // 8, h3, w3_, 1, N.ptr(g2), st), 12 * g2.numel()))
            g1 = conv_bwd2(g2, w2, sc2, q1, 2)
            gskip = None
            if g_f1 is not None:            # the decoder's skip gradient, on the same window: added in K15's epilogue
                gskip = torch.empty((B, 64, 2 * h3, 2 * w3_), device=dev, dtype=torch.float32)
                N.check(_timed("roi_crop", lambda: lib.dmh_roi_crop(N.ptr(_c(g_f1)), None, None, N.ptr(org["h3in"]), B, 64, H // 4,
                                                                   W // 4, 2 * h3, 2 * w3_, 0, N.ptr(gskip), st),
                               8 * gskip.numel()))
            w3ts = frozen_memo(("down_w3ts", w3.data_ptr(), w3._version, sc1.data_ptr()),
                               lambda: _c((w3 * sc1.view(-1, 1, 1, 1)).transpose(0, 1)))
            wdts = frozen_memo(("down_wdts", wd.data_ptr(), wd._version, scd.data_ptr()),
                               lambda: _c((wd * scd.view(-1, 1, 1, 1)).reshape(Co, 64).t()))
            g_f1 = torch.empty((B, 64, 2 * h3, 2 * w3_), device=dev, dtype=torch.float32)
            N.check(_timed("down_conv_bwd", lambda: _k15_bwd(lib, g1, g2, w3ts, wdts, gskip, B, 64, Co, 2 * h3, 2 * w3_, g_f1, st),
                           4 * (g_f1.numel() * (1 if gskip is None else 2) + 2 * g1.numel()), 20 * 64 * g1.numel()))
            f1_frame, f1_org = (2 * h3, 2 * w3_), org["hl_rel"]             # ... or that window, "hl" at its relative origin
        if g_f1 is not None:
            def conv_bwd(g, w, s, res, flag):
                y = torch.empty_like(g)
                N.check(_timed("wino_conv3x3", lambda: lib.dmh_wino_conv3x3_act(
                    N.ptr(g), N.ptr(_wino_filter(w, True, s)), None, N.ptr(res), flag, B, 64, 64, hl, wl, 1, N.ptr(y), st),
                    4 * 3 * g.numel(), 18 * 64 * g.numel()))
                return y

            g_f1 = _c(g_f1)
            g2b = torch.empty_like(f1c)                                     # g * [y_b > 0] on the window
            N.check(_timed("roi_crop", lambda: lib.dmh_roi_crop(N.ptr(g_f1), N.ptr(f1c), None, N.ptr(f1_org), B, 64, f1_frame[0],
                                                               f1_frame[1], hl, wl, 1, N.ptr(g2b), st), 12 * g2b.numel()))
            g1b = conv_bwd(g2b, w2b, s2b, o1b, 2)                           # ... * [out1_b > 0]
            g_ya = conv_bwd(g1b, w1b, s1b, g2b, 0)                          # + the identity branch
            g2a = torch.empty_like(g_ya)                                    # * [y_a > 0]
            N.check(_timed("roi_crop", lambda: lib.dmh_roi_crop(None, N.ptr(ya), N.ptr(g_ya), N.ptr(org["hl"]), B, 64, H // 4,
                                                               W // 4, hl, wl, 1, N.ptr(g2a), st), 12 * g2a.numel()))
            g1a = conv_bwd(g2a, w2a, s2a, o1a, 2)
            g_pool = conv_bwd(g1a, w1a, s1a, g2a, 0)
        n_in = 19 + (15 if ctx.l2 else 0)
        if g_pool is None and g_f0 is None:
            return (None,) * n_in
        g_z = torch.empty_like(f0)
        N.check(_timed("stem_bwd", lambda: lib.dmh_stem_bn_relu_pool_bwd(
            N.ptr(f0), N.ptr(arg), N.ptr(None if g_f0 is None else _c(g_f0)), N.ptr(g_pool), N.ptr(s0), B, 64, hz, wz, N.ptr(g_z),
            st), 4 * 4 * g_z.numel()))
        w_s = frozen_memo(("stem_w_over_std", w_stem.data_ptr(), w_stem._version, 0.225),
                          lambda: _c(w_stem.detach() * (1.0 / 0.225)))
        g_x = torch.zeros((B, 3, H, W), device=dev, dtype=torch.float32)
        hd, wd = plan.size["d"]
        N.check(_timed("stem_conv_bwd_win", lambda: lib.dmh_conv7x7s2_bwd_data_win(
            N.ptr(g_z), N.ptr(w_s), N.ptr(org["d"]), N.ptr(org["hz"]), B, 64, 3, H, W, hd, wd, hz, wz, N.ptr(g_x), st),
            4 * (g_z.numel() + B * 3 * hd * wd), 2 * 147 * 64 * B * (hd // 2) * (wd // 2)))
        return (g_x,) + (None,) * (n_in - 1)


def clean_head(x_clean, conv1_weight, aff0, blocks, layer2=None):
    """CleanHead of the clean frames ``x_clean`` [B,3,H,W]: feature 1 (and, with ``layer2`` = (down block, block) as in
    encoder_head_incremental, feature 2) through the whole-frame kernels, no gradient."""
    with torch.no_grad():
        z = stem_conv_norm(x_clean, conv1_weight, 0.45, 0.225)
        _, y = stem_bn_relu_pool(z, aff0[0], aff0[1])
        for w1, a1, w2, a2 in blocks:
            o = conv3x3_bn_act(y, w1, a1[0], a1[1], None, True, 1)
            y = conv3x3_bn_act(o, w2, a2[0], a2[1], y, True, 1)
        f2 = None
        if layer2 is not None:
            (w3, a1, wd, ad, w2, a2), (v1, c1, v2, c2) = layer2
            f2 = down_block_eval(y, w3, a1[0], a1[1], wd, ad[0], ad[1], w2, a2[0], a2[1])
            o = conv3x3_bn_act(f2, v1, c1[0], c1[1], None, True, 1)
            f2 = conv3x3_bn_act(o, v2, c2[0], c2[1], f2, True, 1)
    return CleanHead(y, f2, x_clean)


def layer2_incremental_ok(x, layer2):
    """layer2 = a 64 -> 128 down-sampling block + a stride-1 block whose kernels (K15, K10) take the whole-frame shape."""
    if layer2 is None:
        return False
    (w3, _, wd, _, w2, _), (v1, _, v2, _) = layer2
    B, _, H, W = x.shape
    probe = x.new_empty((B, 64, H // 4, W // 4))
    return (tuple(w3.shape) == (128, 64, 3, 3) and tuple(wd.shape) == (128, 64, 1, 1) and down_convs_ok(probe, w3, wd)
            and all(tuple(w.shape) == (128, 128, 3, 3) for w in (w2, v1, v2)) and DOWN_NODE_ENABLED
            and _wino_ok(B, 128, 128, H // 8, W // 8, allow_split=False))


def encoder_head_incremental(x, plan, tab, clean, conv1_weight, aff0, blocks, layer2=None):
    """(feature 0 on its "hz" window, feature 1[, feature 2]) of the ResNet-18 encoder for an attack step whose image ``x``
    differs from the clean frames behind ``clean`` (ops.clean_head) inside the plan's boxes only -- see _EncHeadInc.  Marks the
    plan: head_windowed (feature 0's gradient is read inside the window) and f0_compact (feature 0 IS the window).  ``layer2``:
    ((w3, aff1, wd, aff_d, w2, aff2), (w1, aff1, w2, aff2)) -- then layer2 runs on the plan's "h3" window as well."""
    ws = [(b[0], b[2]) for b in blocks]
    if not encoder_head_ok(x, conv1_weight, ws) or not plan.head_incremental_ok:
        raise RuntimeError("encoder_head_incremental: needs ops.frozen_weights(), the ResNet-18 head and a plan whose \"hz\" "
                           "window holds what is read of feature 0")
    if tab.dtype != torch.int32 or tuple(tab.shape) != (len(_ROI_NAMES), x.shape[0], 2) or plan.B != x.shape[0] or \
            (plan.H, plan.W) != tuple(x.shape[2:]) or tuple(clean.work.shape) != (x.shape[0], 64, x.shape[2] // 4, x.shape[3] // 4):
        raise RuntimeError("encoder_head_incremental: plan / origin table / clean feature do not match the image batch")
    d = lambda t: _c(t.detach())        # noqa: E731
    args = [d(conv1_weight), d(aff0[0]), d(aff0[1])]
    for w1, a1, w2, a2 in blocks:
        args += [w1.detach(), d(a1[0]), d(a1[1]), w2.detach(), d(a2[0]), d(a2[1])]
    if layer2 is not None:
        if not (layer2_incremental_ok(x, layer2) and plan.layer2_incremental_ok and clean.work2 is not None):
            raise RuntimeError("encoder_head_incremental: layer2 given, but its shapes / the plan's \"h3\" window / the clean "
                               "feature 2 do not allow the incremental form")
        (w3, a1, wd, ad, w2, a2), (v1, c1, v2, c2) = layer2
        args += [w3.detach(), d(a1[0]), d(a1[1]), wd.detach(), d(ad[0]), d(ad[1]), w2.detach(), d(a2[0]), d(a2[1]),
                 v1.detach(), d(c1[0]), d(c1[1]), v2.detach(), d(c2[0]), d(c2[1])]
    plan.head_windowed = plan.f0_compact = True
    return _EncHeadInc.apply(_c(x), plan, _c(tab), clean, *args)


def masked_depth_errors(disp_gt, disp_pred, mask=None, min_depth=0.1, max_depth=100.0, scale=5.4, clamp_lo=1e-3,
                        clamp_hi=80.0):
    """The eight attack-evaluation metrics of MD2/evaluate_depth.py:57-99 computed from two disparity maps in one
    pass on the device: returns a tensor [abs_err, abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3]."""
    lib = N.lib()
    disp_gt, disp_pred = _c(disp_gt.detach()), _c(disp_pred.detach())
    n = disp_gt.numel()
    if disp_pred.numel() != n or (mask is not None and mask.numel() != n):
        raise RuntimeError("masked_depth_errors: size mismatch")
    part = torch.empty(lib.dmh_depth_errors_partials_size(n), device=disp_gt.device, dtype=torch.float32)
    out = torch.empty(8, device=disp_gt.device, dtype=torch.float32)
    N.check(lib.dmh_masked_depth_errors(N.ptr(disp_gt), N.ptr(disp_pred), N.ptr(None if mask is None else _c(mask)), n,
                                        min_depth, max_depth, scale, clamp_lo, clamp_hi, N.ptr(part), N.ptr(out),
                                        N.stream()))
    return out
