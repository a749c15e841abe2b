"""In-training attack evaluation: the reference's ``MD2/evaluate_depth.py`` pieces that ``Trainer.val`` reaches
(``trainer.py:454-465``): ``compute_errors`` :57-99 and ``evaluate_attacks`` :113-214, with the metric pass fused on the
device (K8, ops.masked_depth_errors) instead of D2H copies + eight numpy reductions per batch."""
import numpy as np
import torch

from . import ops
from .datasets import make_object
from .torchattacks import PGD_depth, Phy_obj_atk, Phy_obj_atk_l0

STEREO_SCALE_FACTOR = 5.4
MIN_DEPTH = 1e-3
MAX_DEPTH = 80


def compute_errors(gt, pred, mask=None):
    """Error metrics between predicted and ground-truth depths (numpy arrays), reference semantics."""
    gt, pred = np.asarray(gt, dtype=np.float64), np.asarray(pred, dtype=np.float64)
    w = np.ones_like(gt) if mask is None else np.asarray(mask, dtype=np.float64)
    assert w.shape == gt.shape and w.shape == pred.shape
    total = w.sum()
    thresh = np.maximum(gt / pred, pred / gt)
    a1, a2, a3 = [((thresh < 1.25 ** k) * w).sum() / total for k in (1, 2, 3)]
    d = gt - pred
    return ((np.abs(d) * w).sum() / total, (np.abs(d) / gt * w).sum() / total, (d ** 2 / gt * w).sum() / total,
            np.sqrt((d ** 2 * w).sum() / total), np.sqrt(((np.log(gt) - np.log(pred)) ** 2 * w).sum() / total), a1, a2, a3)


def evaluate_attacks(model2atk, args, eval_count=25, scene_source=None):
    """Attack the model on ``eval_count`` scene batches and report the mean of the eight metrics between the
    benign and the attacked depth (object-masked for the object attacks).  ``scene_source(n)`` yields n scenes
    [n,3,375,1242]; the KITTI-object loader of the reference (:157-170, starting at index 42) is replaced by it."""
    device = next(model2atk.parameters()).device
    obj_tensor, mask_tensor = make_object(device)
    if args['norm_type'] == "l_inf":
        depth_atk = Phy_obj_atk(model2atk, obj_tensor, mask_tensor, eps=args['epsilon'], alpha=args['alpha'],
                                steps=args['step'])
    elif args['norm_type'] == "l_0":
        depth_atk = Phy_obj_atk_l0(model2atk, obj_tensor, mask_tensor, adam_lr=args["adam_lr"], steps=args["step"],
                                   mask_wt=args["mask_wt"], l0_thresh=args["l0_thresh"])
    elif args['norm_type'] == "image":
        depth_atk = PGD_depth(model2atk, eps=args['epsilon'], alpha=args['alpha'], steps=args['step'])
        depth_atk._targeted = True
    else:
        raise NotImplementedError("evaluation-only attack %r is out of scope (SURVEY.md section 2, row 15)" % (args['norm_type'],))
    if scene_source is None:
        from .datasets import SyntheticKITTIDataset
        data = SyntheticKITTIDataset(320, 1024, [0, "s"], 4, 1 << 30, device, seed=17, pool=max(8, args['batch_size']))
        scene_source = data.next_scenes
    errors = []
    for _ in range(eval_count):
        scene_img = scene_source(args['batch_size'])
        if args['norm_type'] == "image":
            adv_images, ben_images = depth_atk(scene_img)
            obj_masks_out = None
        else:
            adv_images, ben_images, obj_masks_out, _ = depth_atk(scene_img, args['batch_size'], eval=True)
        with torch.no_grad():
            disp_gt = model2atk(ben_images)
            disp_atk = model2atk(adv_images)
        errors.append(ops.masked_depth_errors(disp_gt, disp_atk, obj_masks_out, 0.1, 100, STEREO_SCALE_FACTOR,
                                              MIN_DEPTH, MAX_DEPTH))
    errors = torch.stack(errors)
    mean_errors, max_errors = errors.mean(0).cpu().numpy(), errors.max(0)[0].cpu().numpy()
    names = ("abs_err", "abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3")
    print("Mean Error:\n  " + ("{:>8} | " * 8).format(*names))
    print(("&{: 8.3f}  " * 8).format(*mean_errors.tolist()) + "\\\\")
    print("Max Error:\n  " + ("{:>8} | " * 8).format(*names))
    print(("&{: 8.3f}  " * 8).format(*max_errors.tolist()) + "\\\\")
    return mean_errors
