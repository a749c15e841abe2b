"""CPU restatement of the adversarial sample synthesis and the add-on losses around the hot path.

Reference (under /root/reference/DepthNetworks/monodepth2):
  MonoDataset.prep_adv_data     datasets/mono_dataset.py:186-265
  SimSiam.forward               contrastive.py:62-93
  sup_loss / contras_loss       trainer.py:546-577   (--gt_depth branch :551-557)

Plain PyTorch on tensors (the PIL 8-bit round trip of to_pilimage/to_tensor is not restated; the fixture
tests/golden/prep_adv_data.npz was produced with identity stand-ins for both).  Test infrastructure only (see
oracle/__init__.py).  Pinned by tests/golden/{prep_adv_data,simsiam,addon_losses,addon_gt_depth}.npz (oracle/make_goldens.py ran
the reference functions themselves).
"""
import numpy as np
import torch
import torch.nn as nn

from .attack_ref import PhysicalTransRef

ADV_K = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
ADV_K[0, :] *= 1242      # mono_dataset.py:169-174
ADV_K[1, :] *= 375
STEREO_T = np.eye(4, dtype=np.float32)
STEREO_T[0, 3] = -0.54   # mono_dataset.py:112-117 (side "l" hard-coded)


def prep_adv_data(frame0, frame_s, side, do_flip, adv_trans, ben_trans, z0, alpha):
    """frame0 / frame_s: [3,375,1242] frames as get_color returned them (already flipped when do_flip).
    Returns dict(color_aug_0, color_aug_s, color_ben_0, objmask_0) at 375x1242 (mono_dataset.py:186-253).
    side "l": frame 0 is the left image (plain projection), "s" the right one (project_w_trans);
    side "r": the other way round."""
    z0s, als = [z0], [alpha]
    if side == "l":
        obj0, m0, _, _ = adv_trans.project(1, z0s, als, K=ADV_K)                 # :199
        obj_s, m_s, _, _ = ben_trans.project(1, z0s, als, K=ADV_K, T=STEREO_T)   # :202
        obj0b, m0b, _, _ = ben_trans.project(1, z0s, als, K=ADV_K)               # :236
    else:
        obj_s, m_s, _, _ = ben_trans.project(1, z0s, als, K=ADV_K)               # :207
        obj0, m0, _, _ = adv_trans.project(1, z0s, als, K=ADV_K, T=STEREO_T)     # :210
        obj0b, m0b, _, _ = ben_trans.project(1, z0s, als, K=ADV_K, T=STEREO_T)   # :239
    if do_flip:                                                                   # :215-217, :240-241
        obj0, m0, obj_s, m_s, obj0b, m0b = [torch.flip(t, [3]) for t in (obj0, m0, obj_s, m_s, obj0b, m0b)]
    f0, fs = frame0.unsqueeze(0), frame_s.unsqueeze(0)
    return {"color_aug_0": (f0 * (1 - m0) + obj0 * m0)[0], "color_aug_s": (fs * (1 - m_s) + obj_s * m_s)[0],
            "color_ben_0": (f0 * (1 - m0b) + obj0b * m0b)[0], "objmask_0": m0b.expand(-1, 3, -1, -1)[0]}


class SimSiamRef(nn.Module):
    """contrastive.py:6-93 (same construction order, so the same seed gives the same initial weights)."""

    def __init__(self, dim=1000, pred_dim=512):
        super().__init__()
        prev = 512
        self.projector = nn.Sequential(nn.Linear(prev, prev, bias=False), nn.BatchNorm1d(prev), nn.ReLU(inplace=True),
                                       nn.Linear(prev, prev, bias=False), nn.BatchNorm1d(prev), nn.ReLU(inplace=True),
                                       nn.Linear(prev, dim, bias=False), nn.BatchNorm1d(dim, affine=False))
        self.predictor = nn.Sequential(nn.Linear(dim, pred_dim, bias=False), nn.BatchNorm1d(pred_dim),
                                       nn.ReLU(inplace=True), nn.Linear(pred_dim, dim))

    def forward(self, feature1, feature2):
        z1 = feature1[-1].mean((2, 3))
        z2 = feature2[-1].mean((2, 3))
        z1, z2 = self.projector(z1), self.projector(z2)
        p1, p2 = self.predictor(z1), self.predictor(z2)
        z1, z2 = z1.detach(), z2.detach()
        cos = nn.CosineSimilarity()
        return -(cos(p1, z2).mean() + cos(p2, z1).mean()) * 0.5


def addon_losses(gt_model, simsiam, color_ben, disp0, feats_aug, feats_ben, contras_wt=1.0):
    """trainer.py:546-577: (sup_loss, contras_loss, their sum)."""
    with torch.no_grad():
        disp_gt = gt_model(color_ben)
    sup = nn.MSELoss()(disp_gt, disp0)
    con = simsiam(feats_aug, feats_ben) * contras_wt
    return sup, con, sup + con


def sup_loss_gt_depth(gt_model, color_ben, disp0, color_objmask, objdepth, min_depth=0.1, max_depth=100.0):
    """trainer.py:548-557 with --gt_depth: metric depths (disp_to_depth, layers.py:16-25, x 5.4, clamped to [1e-3, 80]); under
    the pasted object's mask the target is the object's known distance, elsewhere the frozen teacher's depth.
    color_objmask [B,3,H,W], objdepth [B,1,1] as the collated dataset hands them over (mono_dataset.py:254-255)."""
    def depth(d):
        min_disp, max_disp = 1 / max_depth, 1 / min_depth
        return torch.clamp(1 / (min_disp + (max_disp - min_disp) * d) * 5.4, 1e-3, 80)
    with torch.no_grad():
        disp_gt = gt_model(color_ben)
    m = color_objmask[:, [0], :, :]
    gt_depth = m * objdepth.unsqueeze(3) + depth(disp_gt) * (1 - m)
    return nn.MSELoss()(gt_depth, depth(disp0))
