"""Stand-alone supervised hardening loops of the reference on the HIP hot path:

  simple_adv_training.py:96-155   (-at object | object_l0 | image)
  physical_adv_training.py:66-116 (PGD_depth, eps 0.03, alpha 2/255, 10 steps -- `-at image` here)

Per batch: attack -> frozen model's disparity of the benign scenes -> MSE against the robust model's disparity
of the adversarial scenes -> Adam.  Same CLI flags as the reference's root ``options.py:3-18``; KITTI-object
scenes are replaced by the synthetic 375x1242 frames (BASELINE: synthetic data), everything else is the
reference's loop.  ``python -m depthmodelhardening_amd.simple_adv_training -at object -lp note --max_steps 4``
"""
import argparse
import copy
import random
import time

import numpy as np
import torch

from . import ops
from .datasets import SyntheticKITTIDataset, make_object
from .depth_model import import_depth_model
from .my_utils import get_mean_depth_diff
from .torchattacks import PGD_depth, Phy_obj_atk, Phy_obj_atk_l0


def getCLIOptions(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-eps", "--epsilon", default=0.03, type=float, help='norm threshold epsilon')
    ap.add_argument("-alp", "--alpha", default=2 / 255, type=float, help='PGD update weight, alpha')
    ap.add_argument("-s", "--step", default=10, type=int, help='PGD update steps')
    ap.add_argument("-ep", "--epoch", default=20, type=int, help='Total epoches')
    ap.add_argument("-bs", "--batch-size", default=6, type=int, help='training batch size')
    ap.add_argument("-seed", "--random-seed", type=int, default=17, help="random seed in optimization")
    ap.add_argument("-at", "--adv-type", type=str, required=True, choices=['object', 'image', 'object_l0'])
    ap.add_argument("-lp", "--log-postfix", type=str, required=True, help='Log postfix as notes')
    ap.add_argument("--adam_lr", default=0.5, type=float)
    ap.add_argument("--mask_wt", default=0.06, type=float)
    ap.add_argument("--l0_thresh", default=0.1, type=float)
    ap.add_argument("--steps_per_epoch", default=16, type=int, help="NEW: synthetic batches per epoch")
    ap.add_argument("--max_steps", default=0, type=int, help="NEW: stop after this many iterations")
    return vars(ap.parse_args(argv))


def setup_seed(seed):
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def get_atk_model(model_rob, args, device):
    if args['adv_type'] == 'image':
        depth_atk = PGD_depth(model_rob, eps=args['epsilon'], alpha=args['alpha'], steps=args['step'])
        depth_atk._targeted = True
        return depth_atk
    obj_tensor, mask_tensor = make_object(device)
    if args['adv_type'] == 'object':
        return Phy_obj_atk(model_rob, obj_tensor, mask_tensor, eps=args['epsilon'], alpha=args['alpha'],
                           steps=args['step'])
    return Phy_obj_atk_l0(model_rob, obj_tensor, mask_tensor, adam_lr=args['adam_lr'], steps=args['step'],
                          mask_wt=args['mask_wt'], l0_thresh=args['l0_thresh'])


def attack_batch(depth_atk, scene_img, args, eval=False):
    if args['adv_type'] == 'image':
        adv, ben = depth_atk(scene_img)
        return adv, ben, None
    adv, ben, masks, _ = depth_atk(scene_img, args['batch_size'], eval=eval)
    return adv, ben, masks


def do_adv_training(model_rob, model, args, device):
    model_ori = model
    model_ori.eval()
    data = SyntheticKITTIDataset(320, 1024, [0, "s"], 4, 1 << 30, device, seed=args['random_seed'])
    optimizer = torch.optim.Adam(model_rob.parameters(), lr=0.0001)
    depth_atk = get_atk_model(model_rob, args, device)
    step, t0 = 0, time.time()
    for epoch in range(args['epoch']):
        model_rob.train()
        for i in range(args['steps_per_epoch']):
            scene_img_ori = data.next_scenes(args['batch_size'])
            adv_images, ben_images, _ = attack_batch(depth_atk, scene_img_ori, args)
            with torch.no_grad():
                disp_gt = model_ori(ben_images)
            pre_disp = model_rob(adv_images)
            loss = ops.masked_sq_mean(disp_gt - pre_disp, None)      # MSELoss(disp_gt, pre_disp)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            step += 1
            if i % 30 == 0:
                print("epoch %d step %d loss %.6f (%.2f scenes/s)" % (epoch, i, float(loss.detach()),
                                                                     step * args['batch_size'] / (time.time() - t0)))
            if args['max_steps'] and step >= args['max_steps']:
                return model_rob
        model_rob.eval()
        scene = data.next_scenes(args['batch_size'])
        adv_images, ben_images, masks = attack_batch(depth_atk, scene, args, eval=True)
        with torch.no_grad():
            disp_gt, disp_pre, disp_atk = model_ori(ben_images), model_rob(ben_images), model_rob(adv_images)
        print("Performance: model perf: %.4f, attack perf: %.4f" % (
            float(get_mean_depth_diff(disp_pre, disp_gt, None, use_abs=True)),
            float(get_mean_depth_diff(disp_atk, disp_gt, masks, use_abs=True))))
    return model_rob


def main(argv=None):
    args = getCLIOptions(argv)
    setup_seed(args['random_seed'])
    device = torch.device("cuda")
    model = import_depth_model((1024, 320)).to(device).eval()
    model_rob = copy.deepcopy(model).to(device)
    do_adv_training(model_rob, model, args, device)


if __name__ == "__main__":
    main()
