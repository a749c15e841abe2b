"""Globals of the reference's root ``my_utils.py`` that are part of the hot-path boundary.

Reference: my_utils.py:10-14 (device0, object_dataset_root, ori_H, ori_W, train_dist_range),
:19-41 (disp_to_depth, get_mean_depth_diff).
"""
import os

import numpy as np
import torch

# my_utils.py:11 hard-codes the authors' machine; here it is overridable and may not exist at all
object_dataset_root = os.environ.get("DMH_KITTI_OBJECT_ROOT", "/data3/share/kitti/object/")
ori_H = 375
ori_W = 1242
train_dist_range = list(np.arange(5, 10, 0.2))

# P2 of KITTI-object calib 003086 (first two rows are quoted in physicalTrans.py:208-213): used when
# ``{object_dataset_root}/training/calib/003086.txt`` is not on disk (synthetic-data runs).
KITTI_003086_P2 = np.array([[7.215377e+02, 0.0, 6.095593e+02, 4.485728e+01],
                            [0.0, 7.215377e+02, 1.728540e+02, 2.163791e-01],
                            [0.0, 0.0, 1.0, 2.745884e-03]], dtype=np.float64)


def disp_to_depth(disp, min_depth, max_depth):
    """my_utils.py:19-29 (same as MD2/layers.py:16-25)."""
    min_disp = 1 / max_depth
    max_disp = 1 / min_depth
    scaled_disp = min_disp + (max_disp - min_disp) * disp
    depth = 1 / scaled_disp
    return scaled_disp, depth


def get_mean_depth_diff(adv_disp1, ben_disp2, scene_car_mask=None, use_abs=False):
    """my_utils.py:31-41."""
    scaler = 5.4
    if scene_car_mask is None:
        scene_car_mask = torch.ones_like(adv_disp1)
    dep1_adv = torch.clamp(disp_to_depth(torch.abs(adv_disp1), 0.1, 100)[1] * scene_car_mask * scaler, max=100)
    dep2_ben = torch.clamp(disp_to_depth(torch.abs(ben_disp2), 0.1, 100)[1] * scene_car_mask * scaler, max=100)
    if use_abs:
        return torch.sum(torch.abs(dep1_adv - dep2_ben)) / torch.sum(scene_car_mask)
    return torch.sum(dep1_adv - dep2_ben) / torch.sum(scene_car_mask)


def to_device_async(data, device, dtype=None):
    """Small host array / list -> device tensor WITHOUT a host synchronisation: staged through pinned memory and
    copied asynchronously on the current stream.  A pageable `tensor.to(device)` makes the host wait until the GPU has
    drained its queue; the GPU then idles while the host prepares the next launches (measured: five such stalls of
    ~1.8 ms per training step)."""
    import numpy as np
    import torch
    t = torch.from_numpy(np.ascontiguousarray(data)) if isinstance(data, np.ndarray) else torch.as_tensor(data)
    if dtype is not None:
        t = t.to(dtype)
    device = torch.device(device)
    if device.type == "cuda":
        return t.pin_memory().to(device, non_blocking=True)
    return t.to(device)
