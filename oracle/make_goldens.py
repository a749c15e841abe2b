#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE SOURCE ITSELF on the CPU.

Runs only in the build container (needs /root/reference); the GPU box never runs
this.  Nothing from the reference is copied: the reference modules are imported from
where they lie and called; only their numeric inputs/outputs are saved.

Import recipe: packages that are simply not installed here
(torchvision, tensorboardX, cv2, skimage, a few removed numpy/scipy internals) are
replaced by import-time stand-ins.  The only stand-ins that carry arithmetic are the
three torchvision-0.8.2 tensor ops (Resize / Pad / functional.perspective), which
come from oracle/tv082.py ("parity unpinned", see there); everything else the
goldens exercise is the reference's own code: layers.py, trainer.py loss methods,
physicalTrans.py, torchattacks/attacks/{phy_obj_atk,phy_obj_atk_l0,pgd_depth}.py.

Usage:  python oracle/make_goldens.py            (writes tests/golden/)
"""
import os
import random
import sys
import tempfile
import types
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
MD2 = os.path.join(REF, "DepthNetworks", "monodepth2")
DH = os.path.join(REF, "DepthNetworks", "depth-hints")
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from oracle import tv082  # noqa: E402
from oracle.synth import (TinyDepthNet, kitti_like, make_intrinsics, make_object, make_loss_case,  # noqa: E402
                          KITTI_CALIB_TEXT, gt_depth_case, decoder_case)


# --------------------------------------------------------------------------- shims
class _Stub(types.ModuleType):
    """A module whose every missing attribute is a harmless placeholder class."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        obj = type(name, (object,), {"__init__": lambda self, *a, **k: None,
                                     "__call__": lambda self, x, *a, **k: x})
        setattr(self, name, obj)
        return obj


def _stub(name, **attrs):
    m = _Stub(name)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Resize(object):
    def __init__(self, size, interpolation=2):
        self.size = size

    def __call__(self, img):
        return tv082.resize(img, self.size)


class _Pad(object):
    def __init__(self, padding, fill=0, padding_mode="constant"):
        self.padding = padding

    def __call__(self, img):
        return tv082.pad(img, self.padding)


class _ColorJitter(object):
    def __init__(self, *a, **k):
        pass

    @staticmethod
    def get_params(*a, **k):
        return lambda x: x


def install_shims():
    tvf = _stub("torchvision.transforms.functional", perspective=tv082.perspective)
    tvt = _stub("torchvision.transforms", Resize=_Resize, Pad=_Pad, ColorJitter=_ColorJitter, functional=tvf)
    tvm = _stub("torchvision.models")
    tvm.ResNet = type("ResNet", (nn.Module,), {})
    tvm.resnet = _stub("torchvision.models.resnet")
    _stub("torchvision", transforms=tvt, models=tvm)
    _stub("tensorboardX")
    _stub("cv2")
    _stub("skimage")
    _stub("skimage.transform")
    import scipy.optimize
    _stub("scipy.optimize.optimize", _status_message={})
    import scipy.ndimage
    sys.modules.setdefault("scipy.ndimage.filters", scipy.ndimage)
    _stub("numpy.lib.utils")
    _stub("numpy.lib.function_base", flip=np.flip)
    if "numpy.core.numeric" not in sys.modules:
        try:
            import numpy.core.numeric  # noqa: F401
        except Exception:
            _stub("numpy.core.numeric", zeros_like=np.zeros_like)
    # trainer.py:644-645 hard-codes .cuda()
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.argv = ["x"]


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %-28s %7.1f KiB" % (name + ".npz", os.path.getsize(path) / 1024.0))


# --------------------------------------------------------------------------- layers
def gold_layers(layers):
    g = torch.Generator().manual_seed(11)
    x = kitti_like(2, 3, 24, 80, g)
    y = kitti_like(2, 3, 24, 80, g)
    disp = torch.rand(2, 1, 24, 80, generator=g) * 0.9 + 0.05
    K, inv_K = make_intrinsics(2, 24, 80)
    T = torch.eye(4).repeat(2, 1, 1)
    T[0, 0, 3], T[1, 0, 3] = -0.1, 0.1
    scaled, depth = layers.disp_to_depth(disp, 0.1, 100.0)
    bp = layers.BackprojectDepth(2, 24, 80)
    pj = layers.Project3D(2, 24, 80)
    cam = bp(depth, inv_K)
    grid = pj(cam, K, T)
    save("layers_small", x=x, y=y, disp=disp, K=K, inv_K=inv_K, T=T,
         ssim=layers.SSIM()(x, y), smooth=layers.get_smooth_loss(disp, x),
         scaled_disp=scaled, depth=depth, cam_points=cam, grid=grid)


# --------------------------------------------------------------------------- losses
def _fake_trainer(trainer_mod, layers, B, H, W, module_name, use_depth_hints=False, v1_multiscale=False):
    opt = SimpleNamespace(scales=[0, 1, 2, 3], v1_multiscale=v1_multiscale, height=H, width=W, min_depth=0.1,
                          max_depth=100.0, frame_ids=[0, "s"], pose_model_type="separate_resnet",
                          disable_automasking=False, no_ssim=False, adv_train=False, supervised_adv=False,
                          contrastive_learning=False, no_original_train=False, avg_reprojection=v1_multiscale,
                          predictive_mask=False, disparity_smoothness=1e-3, use_depth_hints=use_depth_hints)
    self = SimpleNamespace(opt=opt, num_scales=4, ssim=layers.SSIM(), backproject_depth={}, project_3d={})
    for s in opt.scales:
        self.backproject_depth[s] = layers.BackprojectDepth(B, H // 2 ** s, W // 2 ** s)
        self.project_3d[s] = layers.Project3D(B, H // 2 ** s, W // 2 ** s)
    T = trainer_mod.Trainer
    self.compute_reprojection_loss = lambda pred, target: T.compute_reprojection_loss(self, pred, target)
    if hasattr(T, "compute_loss_masks"):
        self.compute_loss_masks = T.compute_loss_masks
        self.compute_proxy_supervised_loss = T.compute_proxy_supervised_loss
    return self, T


def _run_loss(trainer_mod, layers, case, noise, variant, use_depth_hints=False, v1_multiscale=False):
    """Run the reference generate_images_pred + compute_losses + backward.  ``noise`` is a list of four
    standard-normal tensors (one per scale) that the patched torch.randn hands out, or None -> zeros."""
    inputs, disps = case
    B, _, H, W = inputs[("color", 0, 0)].shape
    self, T = _fake_trainer(trainer_mod, layers, B, H, W, variant, use_depth_hints, v1_multiscale)
    outputs = {}
    leaves = []
    for s, d in enumerate(disps):
        d = d.clone().requires_grad_(True)
        leaves.append(d)
        outputs[("disp", s)] = d
    queue = list(noise) if noise is not None else None
    real_randn = torch.randn

    def fake_randn(*shape, **kw):
        shp = shape[0] if len(shape) == 1 and not isinstance(shape[0], int) else shape
        if queue is None:
            return torch.zeros(*shp)
        t = queue.pop(0)
        assert tuple(t.shape) == tuple(shp), (t.shape, shp)
        return t

    torch.randn = fake_randn
    try:
        T.generate_images_pred(self, inputs, outputs)
        losses = T.compute_losses(self, inputs, outputs)
    finally:
        torch.randn = real_randn
    losses["loss"].backward()
    res = {"loss": losses["loss"]}
    for s in range(4):
        res["loss_%d" % s] = losses["loss/%d" % s]
        res["identity_selection_%d" % s] = outputs["identity_selection/%d" % s]
        res["grad_disp_%d" % s] = leaves[s].grad
        res["warped_%d" % s] = outputs[("color", "s", s)]
        if "reproj_loss/%d" % s in losses:
            res["reproj_loss_%d" % s] = losses["reproj_loss/%d" % s]
        if "depth_hint_loss/%d" % s in losses:
            res["depth_hint_loss_%d" % s] = losses["depth_hint_loss/%d" % s]
            res["depth_hint_pixels_%d" % s] = outputs["depth_hint_pixels/%d" % s]
    if ("color_depth_hint", "s", 0) in outputs:
        res["warped_hint"] = outputs[("color_depth_hint", "s", 0)]
    res["depth_0"] = outputs[("depth", 0, 0)]
    res["sample_0"] = outputs[("sample", "s", 0)]
    return res


def gold_losses(trainer_mod, layers, tag, only=None):
    shapes = {"small": (2, 32, 96, 21), "cfg1": (2, 192, 640, 22)}
    if tag == "md2":
        shapes["hd"] = (2, 320, 1024, 23)       # the headline resolution (BASELINE configs 2-5)
    for name, (B, H, W, seed) in shapes.items():
        if only is not None and name not in only:
            continue
        case = make_loss_case(B, H, W, seed)
        g = torch.Generator().manual_seed(seed + 100)
        nshape = (B, 1, H, W)
        noise = [torch.randn(*nshape, generator=g) for _ in range(4)]
        res0 = _run_loss(trainer_mod, layers, case, None, tag)
        res1 = _run_loss(trainer_mod, layers, case, noise, tag)
        keep = {"shape": np.array([B, H, W, seed])}
        # inputs are regenerated from the seed by oracle.synth.make_loss_case; only outputs are stored
        big, hd = name != "small", name == "hd"
        for k, v in res0.items():
            if big and (k.startswith("warped") or k.startswith("sample") or k.startswith("depth")):
                continue
            keep["nonoise_" + k] = v
        for k, v in res1.items():
            if k.startswith("warped") or k.startswith("sample") or k.startswith("depth"):
                continue
            if hd and not k.startswith("loss"):     # 320 x 1024: the noise run keeps its loss values only (file size)
                continue
            keep["noise_" + k] = v
        for k in list(keep):
            if "identity_selection" in k:      # 0/1 maps: store as bits
                keep[k] = np.packbits(keep[k].numpy().astype(np.uint8))
            elif big and (k.endswith("grad_disp_0") or (hd and "grad_disp_" in k and not k.endswith("_3"))):
                # every 3rd row/col + exact double sums
                g0 = keep.pop(k)
                keep[k + "_sub3"] = g0[:, :, ::3, ::3]
                keep[k + "_sum"] = g0.double().sum((1, 2, 3))
                keep[k + "_abssum"] = g0.double().abs().sum((1, 2, 3))
        save("loss_%s_%s" % (tag, name), **keep)


def gold_v1_multiscale(trainer_mod, layers):
    """MD2 with --v1_multiscale --avg_reprojection (trainer.py:478-483,593-596,617-621): per-scale warps at the scale's
    own resolution and intrinsics; with the one stereo source frame the average over frames is the frame itself."""
    from oracle.synth import add_pyramid
    B, H, W, seed = 2, 64, 192, 27
    inputs, disps = make_loss_case(B, H, W, seed)
    add_pyramid(inputs, B, H, W)
    g = torch.Generator().manual_seed(seed + 100)
    noise = [torch.randn(B, 1, H >> s, W >> s, generator=g) for s in range(4)]
    keep = {"shape": np.array([B, H, W, seed])}
    for tag, nz in (("nonoise", None), ("noise", noise)):
        res = _run_loss(trainer_mod, layers, (inputs, disps), nz, "md2", v1_multiscale=True)
        for k, v in res.items():
            if k.startswith(("sample", "depth_0", "warped")):
                continue
            if "identity_selection" in k:
                v = np.packbits(v.numpy().astype(np.uint8))
            keep[tag + "_" + k] = v
    save("loss_md2_v1ms", **keep)


def gold_options(trainer_mod, layers, tag="md2"):
    """The option branches of the per-scale loss body (MD2 trainer.py:608-658; tag "dh": depth-hints/trainer.py:638-741, which
    reduces over the frames first and normalises by its masks): --predictive_mask (with --disable_automasking, one and two source
    frames) and --avg_reprojection over two source frames (auto-masking on, recorded tie-break noise); DepthHints also with
    --use_depth_hints beside --avg_reprojection (frames -1 and "s": the hint is warped with the stereo pose)."""
    from oracle.synth import options_case, make_depth_hint
    B, H, W, seed = 2, 32, 96, 33
    cases = [("pmask", ["s"], dict(disable_automasking=True, predictive_mask=True)),
             ("pmask2", [-1, "s"], dict(disable_automasking=True, predictive_mask=True)),
             ("avg2", [-1, 1], dict(avg_reprojection=True)),
             ("avg2_noauto", [-1, 1], dict(avg_reprojection=True, disable_automasking=True))]
    if tag == "dh":
        cases.append(("avg2s_hints", [-1, "s"], dict(avg_reprojection=True, use_depth_hints=True)))
    for name, frames, kw in cases:
        inputs, disps, poses, masks = options_case(B, H, W, seed, frames)
        if kw.get("use_depth_hints"):
            inputs["depth_hint"], inputs["depth_hint_mask"] = make_depth_hint(B, H, W, seed + 50)
        self, T = _fake_trainer(trainer_mod, layers, B, H, W, tag)
        self.opt.frame_ids = [0] + frames
        for k, v in kw.items():
            setattr(self.opt, k, v)
        outputs, leaves, mleaves = {}, [], []
        for f, P in poses.items():
            outputs[("cam_T_cam", 0, f)] = P
        for s, d in enumerate(disps):
            d = d.clone().requires_grad_(True)
            leaves.append(d)
            outputs[("disp", s)] = d
        if self.opt.predictive_mask:
            outputs["predictive_mask"] = {}
            for s, m in enumerate(masks):
                m = m.clone().requires_grad_(True)
                mleaves.append(m)
                outputs["predictive_mask"][("disp", s)] = m
        g = torch.Generator().manual_seed(seed + 100)
        queue = [torch.randn(B, 1, H, W, generator=g) for _ in range(4)]       # avg: one identity channel
        real_randn = torch.randn

        def fake_randn(*shape, **k2):
            shp = shape[0] if len(shape) == 1 and not isinstance(shape[0], int) else shape
            t = queue.pop(0)
            assert tuple(t.shape) == tuple(shp), (t.shape, shp)
            return t
        torch.randn = fake_randn
        try:
            T.generate_images_pred(self, inputs, outputs)
            losses = T.compute_losses(self, inputs, outputs)
        finally:
            torch.randn = real_randn
        losses["loss"].backward()
        keep = {"shape": np.array([B, H, W, seed]), "loss": losses["loss"]}
        for s in range(4):
            keep["loss_%d" % s] = losses["loss/%d" % s]
            keep["grad_disp_%d" % s] = leaves[s].grad
            if mleaves:
                keep["grad_mask_%d" % s] = mleaves[s].grad
            if "identity_selection/%d" % s in outputs:
                keep["identity_selection_%d" % s] = np.packbits(outputs["identity_selection/%d" % s].numpy().astype(np.uint8))
            for k in ("reproj_loss", "depth_hint_loss"):
                if "%s/%d" % (k, s) in losses:
                    keep["%s_%d" % (k, s)] = losses["%s/%d" % (k, s)]
            if "depth_hint_pixels/%d" % s in outputs:
                keep["depth_hint_pixels_%d" % s] = np.packbits(outputs["depth_hint_pixels/%d" % s].numpy().astype(np.uint8))
        save("loss_%s_opt_%s" % (tag, name), **keep)


def gold_depth_hints(trainer_mod, layers):
    """DepthHints with --use_depth_hints (depth-hints/trainer.py:510-525,629-636,700-725): hint warp, three-way argmin,
    proxy log-L1 supervision; synthetic hints with holes (oracle.synth.make_depth_hint)."""
    from oracle.synth import make_depth_hint
    for name, (B, H, W, seed) in {"small": (2, 32, 96, 21), "cfg1": (2, 192, 640, 22)}.items():
        inputs, disps = make_loss_case(B, H, W, seed)
        inputs["depth_hint"], inputs["depth_hint_mask"] = make_depth_hint(B, H, W, seed + 50)
        g = torch.Generator().manual_seed(seed + 100)
        noise = [torch.randn(B, 1, H, W, generator=g) for _ in range(4)]
        keep = {"shape": np.array([B, H, W, seed])}
        for tag, nz in (("nonoise", None), ("noise", noise)):
            res = _run_loss(trainer_mod, layers, (inputs, disps), nz, "dh", use_depth_hints=True)
            for k, v in res.items():
                if k.startswith(("sample", "depth_0")) or (k.startswith("warped") and (name != "small" or tag != "nonoise")):
                    continue
                if "identity_selection" in k or "depth_hint_pixels" in k:
                    v = np.packbits(v.numpy().astype(np.uint8))
                elif name != "small" and k.startswith("grad_disp"):
                    v = v[:, :, ::3, ::3] if k.endswith("_0") else v
                keep[tag + "_" + k] = v
        save("loss_dh_hints_%s" % name, **keep)


# --------------------------------------------------------------------------- attacks
def _seed_all(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def gold_geometry(PhysicalTrans, calib_path):
    obj, mask = make_object()
    pt = PhysicalTrans(obj, mask, {"path": calib_path}, (1, 3, 375, 1242), dist_range=list(np.arange(5, 10, 0.2)))
    quads = np.zeros((25, 13, 4, 2), dtype=np.int32)
    for i, z0 in enumerate(pt.dist_range):
        for j, al in enumerate(pt.angle_range):
            quads[i, j] = pt.objPosOnImage(z0, al)
    adv_K = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
    adv_K[0, :] *= 1242
    adv_K[1, :] *= 375
    quads_K = np.zeros((25, 13, 4, 2), dtype=np.int32)
    for i, z0 in enumerate(pt.dist_range):
        for j, al in enumerate(pt.angle_range):
            quads_K[i, j] = pt.objPosOnImage(z0, al, adv_K)
    # one full-size paste: the reference's project() on fixed samples
    o, m, _, _ = pt.project(batch_size=2, z0_sample=[5.0, 9.4], alpha_sample=[-30, 15])
    save("geometry", quads=quads, quads_K=quads_K, start=np.array(pt.pos_obj_img_start, dtype=np.int32),
         proj_img_rows=o[:, :, 150:260:11, 400:900:7], proj_mask_rows=m[:, :, 150:260:11, 400:900:7],
         proj_img_sum=o.double().sum((2, 3)), proj_mask_sum=m.double().sum((2, 3)))


def ref_cal_l0(atk0):
    return np.array(int(atk0.cal_l0()))


def gold_attacks(ta):
    obj, mask = make_object()
    g = torch.Generator().manual_seed(31)
    scenes = kitti_like(2, 3, 375, 1242, g)
    model = TinyDepthNet(seed=5)

    # --- Phy_obj_atk, 3 steps, Ba=2 (BASELINE config 1 attack shape)
    atk = ta.Phy_obj_atk(model, obj, mask, eps=0.1, alpha=0.02, steps=3, dist_range=list(np.arange(5, 10, 0.2)))
    _seed_all(41)
    model.train()
    adv_s, ben_s, m_out, patch = atk(scenes, 2)
    assert model.training
    save("atk_linf", shape=np.array([2, 3, 41]), patch_sub=patch[:, :, ::2, ::2], patch_sum=patch.double().sum(), mask_out_sum=m_out.double().sum((1, 2, 3)),
         adv_sum=adv_s.double().sum((2, 3)), ben_sum=ben_s.double().sum((2, 3)),
         adv_rows=adv_s[:, :, 120:300:9, 300:800:5], ben_rows=ben_s[:, :, 120:300:9, 300:800:5],
         mask_rows=m_out[:, :, 120:300:9, 300:800:5])

    # --- Phy_obj_atk_l0, steps=3 (<= 6 iterations), Ba=2
    atk0 = ta.Phy_obj_atk_l0(model, obj, mask, adam_lr=0.5, steps=3, mask_wt=0.06, l0_thresh=0.1,
                             dist_range=list(np.arange(5, 10, 0.2)))
    _seed_all(43)
    adv_s, ben_s, m_out, patch = atk0(scenes, 2)
    save("atk_l0", shape=np.array([2, 3, 43]), patch_sub=patch[:, :, ::2, ::2], patch_sum=patch.double().sum(),
         mask_out_sum=m_out.double().sum((1, 2, 3)),
         adv_sum=adv_s.double().sum((2, 3)), ben_sum=ben_s.double().sum((2, 3)),
         pattern_pos_sub=atk0.pattern_pos_tensor[:, :, ::2, ::2], pattern_neg_sub=atk0.pattern_neg_tensor[:, :, ::2, ::2],
         pattern_pos_sum=atk0.pattern_pos_tensor.double().sum(), pattern_neg_sum=atk0.pattern_neg_tensor.double().sum(),
         l0_final=ref_cal_l0(atk0), final_mask_weight=np.array(float(atk0.mask_weight)))

    # --- PGD_depth, 3 steps, targeted (simple_adv_training.py:40-41) and untargeted
    imgs = kitti_like(2, 3, 320, 1024, torch.Generator().manual_seed(33))
    for targeted in (True, False):
        p = ta.PGD_depth(model, eps=0.03, alpha=2 / 255, steps=3, random_start=True)
        p._targeted = targeted
        _seed_all(47)
        adv, clean = p(imgs)
        save("atk_pgd_%s" % ("targeted" if targeted else "untargeted"), shape=np.array([2, 3, 47]),
             adv_rows=adv[:, :, ::16, ::8], adv_sum=adv.double().sum((2, 3)),
             delta_absmax=(adv - clean).abs().amax())


# --------------------------------------------------------------------------- add-ons (a-14, a-15, f-1, f-2)
def gold_simsiam(contrastive):
    """MD2/contrastive.py:6-93 on fixed features: loss, feature gradients, per-parameter gradient sums (train mode)."""
    torch.manual_seed(51)
    net = contrastive.SimSiam()
    net.train()
    g = torch.Generator().manual_seed(52)
    f_adv = [torch.rand(4, 512, 3, 5, generator=g).requires_grad_(True)]
    f_ben = [(f_adv[0].detach() + 0.3 * torch.rand(4, 512, 3, 5, generator=g)).requires_grad_(True)]
    loss = net(f_adv, f_ben)
    loss.backward()
    names, gsum, wsum = [], [], []
    for n, p in net.named_parameters():
        names.append(n)
        gsum.append(float(p.grad.double().abs().sum()))
        wsum.append(float(p.detach().double().abs().sum()))
    save("simsiam", seeds=np.array([51, 52]), loss=loss.detach(), g_adv=f_adv[0].grad[:, ::8], g_ben=f_ben[0].grad[:, ::8],
         param_names=np.array(names), param_grad_abssum=np.array(gsum), param_abssum=np.array(wsum),
         running_mean_0=net.projector[1].running_mean.clone())


def gold_sup_loss(trainer_mod, layers):
    """MD2/trainer.py:546-577 with --adv_train --supervised_adv --contrastive_learning --no_original_train: the add-on
    losses alone (sup_loss = MSE(gt_model(color_ben), disp_0); contras_loss = SimSiam(features_aug, features_ben))."""
    import contrastive
    B, H, W = 8, 32, 96      # SimSiam's BatchNorm1d needs a real batch: over a batch of 2 every feature becomes +-1 and the
    g = torch.Generator().manual_seed(61)   # feature gradient is ill-conditioned by construction
    color_ben = kitti_like(B, 3, H, W, g)
    disp = (torch.rand(B, 1, H, W, generator=g) * 0.3 + 0.01).requires_grad_(True)
    torch.manual_seed(62)
    simsiam = contrastive.SimSiam()
    simsiam.train()
    feats_aug = [torch.rand(B, 512, 1, 3, generator=g).requires_grad_(True)]
    feats_ben = [torch.rand(B, 512, 1, 3, generator=g).requires_grad_(True)]
    opt = SimpleNamespace(adv_train=True, supervised_adv=True, contrastive_learning=True, no_original_train=True,
                          gt_depth=False, min_depth=0.1, max_depth=100.0)
    self = SimpleNamespace(opt=opt, gt_model=TinyDepthNet(seed=5).eval(), sup_loss_creteria=nn.MSELoss(),
                           models={"contrastive_learning": simsiam})
    inputs = {("color_ben", 0, 0): color_ben}
    outputs = {("disp", 0): disp, "middle_features_aug": feats_aug, "middle_features_ben": feats_ben}
    losses = trainer_mod.Trainer.compute_losses(self, inputs, outputs)
    losses["loss"].backward()
    save("addon_losses", shape=np.array([B, H, W, 61, 62]), sup_loss=losses["sup_loss"].detach(),
         contras_loss=losses["contras_loss"].detach(), loss=losses["loss"].detach(), g_disp=disp.grad,
         g_feat_aug=feats_aug[0].grad)


def gold_gt_depth(trainer_mod, layers):
    """MD2/trainer.py:546-557 with --adv_train --supervised_adv --gt_depth --no_original_train: sup_loss on metric depths."""
    color_ben, disp, mask, objdepth = gt_depth_case()
    disp = disp.requires_grad_(True)
    opt = SimpleNamespace(adv_train=True, supervised_adv=True, contrastive_learning=False, no_original_train=True,
                          gt_depth=True, min_depth=0.1, max_depth=100.0)
    self = SimpleNamespace(opt=opt, gt_model=TinyDepthNet(seed=5).eval(), sup_loss_creteria=nn.MSELoss(), models={})
    inputs = {("color_ben", 0, 0): color_ben, ("color_objmask", 0, 0): mask, ("objdepth", 0, 0): objdepth}
    losses = trainer_mod.Trainer.compute_losses(self, inputs, {("disp", 0): disp})
    losses["loss"].backward()
    save("addon_gt_depth", seed=np.array([63]), sup_loss=losses["sup_loss"].detach(), loss=losses["loss"].detach(),
         g_disp=disp.grad, clamped_frac=np.array([float((disp.grad == 0).float().mean())]))


def gold_unet_decoder(networks):
    """MD2/networks/depth_decoder.py:17-65 -- the reference's own DepthDecoder (with layers.ConvBlock / Conv3x3 / upsample) on
    seeded features: the four disparities, the feature gradients of a weighted sum, and per-parameter checksums of the
    seeded initial weights and of their gradients.  Pins oracle/unet_ref.DepthDecoderRef (same seed -> same weights)."""
    torch.manual_seed(71)
    dec = networks.DepthDecoder(np.array([64, 64, 128, 256, 512]))
    feats, wts = decoder_case()
    feats = [f.requires_grad_(True) for f in feats]
    out = dec(feats)
    total = sum((out[("disp", s)] * wts[s]).sum() for s in range(4))
    total.backward()
    names = [n for n, _ in dec.named_parameters()]
    keep = {"seeds": np.array([71, 72]), "total": total.detach(), "param_names": np.array(names),
            "param_abssum": np.array([float(p.detach().double().abs().sum()) for _, p in dec.named_parameters()]),
            "param_grad_abssum": np.array([float(p.grad.double().abs().sum()) for _, p in dec.named_parameters()])}
    for s in range(4):
        keep["disp_%d" % s] = out[("disp", s)].detach()
    for k, f in enumerate(feats):
        keep["g_feat_%d" % k] = f.grad[:, ::8, ::2, ::2] if k < 2 else f.grad[:, ::8]
    save("unet_decoder", **keep)


def gold_compute_errors(evaluate_depth, layers):
    """MD2/evaluate_depth.py:57-99 both branches, fed through the depth conversion of :193-194."""
    g = torch.Generator().manual_seed(71)
    disp_gt = torch.rand(2, 1, 40, 72, generator=g) * 0.6 + 0.005
    disp_atk = (disp_gt * (0.6 + 0.8 * torch.rand(2, 1, 40, 72, generator=g))).clamp(1e-4, 1.0)
    disp_atk[0, 0, :3] *= -1.0       # the reference takes |disp|
    mask = (torch.rand(2, 1, 40, 72, generator=g) > 0.7).float()
    S, lo, hi = evaluate_depth.STEREO_SCALE_FACTOR, evaluate_depth.MIN_DEPTH, evaluate_depth.MAX_DEPTH
    gt_depth = torch.clamp(layers.disp_to_depth(torch.abs(disp_gt), 0.1, 100)[1] * S, max=hi, min=lo)
    atk_depth = torch.clamp(layers.disp_to_depth(torch.abs(disp_atk), 0.1, 100)[1] * S, max=hi, min=lo)
    e_all = evaluate_depth.compute_errors(gt_depth.numpy(), atk_depth.numpy(), mask=None)
    e_msk = evaluate_depth.compute_errors(gt_depth.numpy(), atk_depth.numpy(), mask=mask.numpy())
    save("compute_errors", seed=np.array(71), disp_gt=disp_gt, disp_atk=disp_atk, mask=mask, gt_depth=gt_depth,
         atk_depth=atk_depth, errors_all=np.array(e_all, dtype=np.float64), errors_masked=np.array(e_msk, dtype=np.float64))


def gold_prep_adv_data(mono_dataset, PhysicalTrans, calib_path):
    """MD2/datasets/mono_dataset.py:186-265 (prep_adv_data) run unbound on tensors (ToTensor / ToPILImage are the
    identity stand-ins, so the PIL 8-bit round trip is not part of the fixture), both camera sides, with and without
    do_flip, torchvision's perspective = oracle/tv082.py ("parity unpinned")."""
    obj, mask = make_object()
    g = torch.Generator().manual_seed(81)
    obj_adv = (obj + 0.1 * (torch.rand(obj.shape, generator=g) - 0.5)).clamp(0, 1)
    raw_l, raw_r = kitti_like(1, 3, 375, 1242, g)[0], kitti_like(1, 3, 375, 1242, g)[0]
    dist_range = list(np.arange(5, 10, 0.2))
    adv_K = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
    adv_K[0, :] *= 1242
    adv_K[1, :] *= 375
    stereo_T = np.eye(4, dtype=np.float32)
    stereo_T[0, 3] = -0.54
    ident = lambda x: x  # noqa: E731
    keep = {"seed": np.array(81)}
    for side in ("l", "r"):
        for do_flip in (False, True):
            self = SimpleNamespace(to_tensor=ident, to_pilimage=ident, ori_H=375, ori_W=1242, resize_trans=ident,
                                   half_no_synthesis=False, adv_K=adv_K, stereo_T=stereo_T,
                                   ben_trans=PhysicalTrans(obj, mask, {"path": calib_path}, (1, 3, 375, 1242), dist_range=dist_range),
                                   adv_trans=PhysicalTrans(obj_adv, mask, {"path": calib_path}, (1, 3, 375, 1242), dist_range=dist_range))
            # get_color has already flipped the frames when do_flip is set (mono_dataset.py:297-304)
            f0, fs = (raw_l, raw_r) if side == "l" else (raw_r, raw_l)
            if do_flip:
                f0, fs = torch.flip(f0, [2]), torch.flip(fs, [2])
            inputs = {("color", 0, -1): f0, ("color", "s", -1): fs}
            _seed_all(90 + (side == "r") * 2 + int(do_flip))
            out = mono_dataset.MonoDataset.prep_adv_data(self, inputs, side, do_flip, load_ben_color=True)
            tag = "%s%d_" % (side, int(do_flip))
            sub = (slice(None), slice(100, 330, 6), slice(300, 1000, 5))
            keep[tag + "z0"] = np.array(float(out[("objdepth", 0, 0)].reshape(-1)[0]))
            keep[tag + "aug0"] = out[("color_aug", 0, -1)][sub]
            keep[tag + "aug_s"] = out[("color_aug", "s", -1)][sub]
            keep[tag + "ben0"] = out[("color_ben", 0, -1)][sub]
            keep[tag + "mask0"] = out[("color_objmask", 0, -1)][sub]
            keep[tag + "aug0_sum"] = out[("color_aug", 0, -1)].double().sum()
            keep[tag + "mask0_sum"] = out[("color_objmask", 0, -1)][0].double().sum()
            assert out[("color", 0, -1)] is out[("color_ben", 0, -1)] and out[("color", "s", -1)] is out[("color_aug", "s", -1)]
    save("prep_adv_data", **keep)


def main():
    only = set(sys.argv[1:])        # e.g. `python oracle/make_goldens.py options`: just that group (default: everything)

    def want(name):
        return not only or name in only
    install_shims()
    tmp = tempfile.mkdtemp(prefix="kitti_obj_")
    os.makedirs(os.path.join(tmp, "training", "calib"))
    calib = os.path.join(tmp, "training", "calib", "003086.txt")
    with open(calib, "w") as f:
        f.write(KITTI_CALIB_TEXT)

    sys.path.insert(0, MD2)
    sys.path.append(REF)
    import my_utils
    my_utils.object_dataset_root = tmp  # bound at import by the attack modules (phy_obj_atk.py:7,53-54)
    import layers
    if want("layers"):
        gold_layers(layers)
    import trainer as md2_trainer
    if want("losses"):
        gold_losses(md2_trainer, layers, "md2")
    elif want("hd"):
        gold_losses(md2_trainer, layers, "md2", only=("hd",))
    if want("v1ms"):
        gold_v1_multiscale(md2_trainer, layers)
    if want("options"):
        gold_options(md2_trainer, layers)
    import physicalTrans
    if want("geometry"):
        gold_geometry(physicalTrans.PhysicalTrans, calib)
    import torchattacks as ta
    if want("attacks"):
        gold_attacks(ta)
    import contrastive
    if want("addons"):
        gold_simsiam(contrastive)
        gold_sup_loss(md2_trainer, layers)
    if want("addons") or want("gt_depth"):
        gold_gt_depth(md2_trainer, layers)
    if want("unet"):
        import networks as md2_networks        # resnet_encoder.py's torchvision import meets the stand-in; only the decoder runs
        gold_unet_decoder(md2_networks)
    import evaluate_depth
    if want("errors"):
        gold_compute_errors(evaluate_depth, layers)
    import datasets.mono_dataset as mono_dataset
    if want("prep"):
        gold_prep_adv_data(mono_dataset, physicalTrans.PhysicalTrans, calib)

    # DepthHints variant: same module names, so import it in a scrubbed namespace
    for m in ["trainer", "layers", "datasets", "networks", "utils", "kitti_utils", "options", "evaluate_depth",
              "contrastive"] + [k for k in sys.modules if k.startswith(("datasets.", "networks."))]:
        sys.modules.pop(m, None)
    sys.path[0] = DH
    import layers as dh_layers
    import trainer as dh_trainer
    if want("losses"):
        gold_losses(dh_trainer, dh_layers, "dh")
    if want("hints"):
        gold_depth_hints(dh_trainer, dh_layers)
    if want("options") or want("options_dh"):
        gold_options(dh_trainer, dh_layers, "dh")


if __name__ == "__main__":
    main()
