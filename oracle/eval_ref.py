"""CPU restatement of the attack-evaluation metrics (MD2/evaluate_depth.py:57-99 compute_errors, :193-197 depth
conversion).  Test infrastructure only (see oracle/__init__.py)."""
import numpy as np


def disp_to_eval_depth(disp, min_depth=0.1, max_depth=100.0, scale=5.4, lo=1e-3, hi=80.0):
    """:193-194: clamp(disp_to_depth(|disp|, 0.1, 100)[1] * STEREO_SCALE_FACTOR, MIN_DEPTH, MAX_DEPTH)."""
    min_disp, max_disp = 1 / max_depth, 1 / min_depth
    depth = 1 / (min_disp + (max_disp - min_disp) * np.abs(disp))
    return np.clip(depth * scale, lo, hi)


def compute_errors(gt, pred, mask=None):
    """:57-99, both branches."""
    if mask is None:
        thresh = np.maximum((gt / pred), (pred / gt))
        a1, a2, a3 = (thresh < 1.25).mean(), (thresh < 1.25 ** 2).mean(), (thresh < 1.25 ** 3).mean()
        abs_err = np.mean(np.abs(gt - pred))
        rmse = np.sqrt(((gt - pred) ** 2).mean())
        rmse_log = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
        abs_rel = np.mean(np.abs(gt - pred) / gt)
        sq_rel = np.mean(((gt - pred) ** 2) / gt)
    else:
        assert mask.shape == gt.shape and mask.shape == pred.shape
        total = mask.sum()
        thresh = np.maximum((gt / pred), (pred / gt))
        a1 = ((thresh < 1.25) * mask).sum() / total
        a2 = ((thresh < 1.25 ** 2) * mask).sum() / total
        a3 = ((thresh < 1.25 ** 3) * mask).sum() / total
        abs_err = (np.abs(gt - pred) * mask).sum() / total
        rmse = np.sqrt((((gt - pred) ** 2) * mask).sum() / total)
        rmse_log = np.sqrt((((np.log(gt) - np.log(pred)) ** 2) * mask).sum() / total)
        abs_rel = np.sum(np.abs(gt - pred) / gt * mask) / total
        sq_rel = np.sum(((gt - pred) ** 2) / gt * mask) / total
    return abs_err, abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3
