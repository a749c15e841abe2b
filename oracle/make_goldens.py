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
                          KITTI_CALIB_TEXT)


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
def _fake_trainer(trainer_mod, layers, B, H, W, module_name):
    opt = SimpleNamespace(scales=[0, 1, 2, 3], v1_multiscale=False, height=H, width=W, min_depth=0.1,
                          max_depth=100.0, frame_ids=[0, "s"], pose_model_type="separate_resnet",
                          disable_automasking=False, no_ssim=False, adv_train=False, supervised_adv=False,
                          contrastive_learning=False, no_original_train=False, avg_reprojection=False,
                          predictive_mask=False, disparity_smoothness=1e-3, use_depth_hints=False)
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


def _run_loss(trainer_mod, layers, case, noise, variant):
    """Run the reference generate_images_pred + compute_losses + backward.  ``noise`` is a list of four
    standard-normal tensors (one per scale) that the patched torch.randn hands out, or None -> zeros."""
    inputs, disps = case
    B, _, H, W = inputs[("color", 0, 0)].shape
    self, T = _fake_trainer(trainer_mod, layers, B, H, W, variant)
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
    res["depth_0"] = outputs[("depth", 0, 0)]
    res["sample_0"] = outputs[("sample", "s", 0)]
    return res


def gold_losses(trainer_mod, layers, tag):
    for name, (B, H, W, seed) in {"small": (2, 32, 96, 21), "cfg1": (2, 192, 640, 22)}.items():
        case = make_loss_case(B, H, W, seed)
        g = torch.Generator().manual_seed(seed + 100)
        nshape = (B, 1, H, W)
        noise = [torch.randn(*nshape, generator=g) for _ in range(4)]
        res0 = _run_loss(trainer_mod, layers, case, None, tag)
        res1 = _run_loss(trainer_mod, layers, case, noise, tag)
        keep = {"shape": np.array([B, H, W, seed])}
        # inputs are regenerated from the seed by oracle.synth.make_loss_case; only outputs are stored
        big = name != "small"
        for k, v in res0.items():
            if big and (k.startswith("warped") or k.startswith("sample") or k.startswith("depth")):
                continue
            keep["nonoise_" + k] = v
        for k, v in res1.items():
            if k.startswith("warped") or k.startswith("sample") or k.startswith("depth"):
                continue
            keep["noise_" + k] = v
        for k in list(keep):
            if "identity_selection" in k:      # 0/1 maps: store as bits
                keep[k] = np.packbits(keep[k].numpy().astype(np.uint8))
            elif big and k.endswith("grad_disp_0"):   # every 3rd row/col + exact double sums
                g0 = keep.pop(k)
                keep[k + "_sub3"] = g0[:, :, ::3, ::3]
                keep[k + "_sum"] = g0.double().sum((1, 2, 3))
                keep[k + "_abssum"] = g0.double().abs().sum((1, 2, 3))
        save("loss_%s_%s" % (tag, name), **keep)


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


def main():
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
    gold_layers(layers)
    import trainer as md2_trainer
    gold_losses(md2_trainer, layers, "md2")
    import physicalTrans
    gold_geometry(physicalTrans.PhysicalTrans, calib)
    import torchattacks as ta
    gold_attacks(ta)

    # DepthHints variant: same module names, so import it in a scrubbed namespace
    for m in ["trainer", "layers", "datasets", "networks", "utils", "kitti_utils", "options", "evaluate_depth",
              "contrastive"] + [k for k in sys.modules if k.startswith(("datasets.", "networks."))]:
        sys.modules.pop(m, None)
    sys.path[0] = DH
    import layers as dh_layers
    import trainer as dh_trainer
    gold_losses(dh_trainer, dh_layers, "dh")


if __name__ == "__main__":
    main()
