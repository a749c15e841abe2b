#!/usr/bin/env python3
"""Which 3x3 convolution shapes of one adversarial-training step take the K10 Winograd kernel and which fall through to
MIOpen: python3 tools/conv_census.py  (counts per (B, C_in, C_out, H_out, W_out) and the dispatch decision)."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthmodelhardening_amd import ops  # noqa: E402

census = collections.Counter()
real_ok = ops._wino_ok


def counting_ok(B, n_in, n_out, Ho, Wo, allow_split=True):
    r = real_ok(B, n_in, n_out, Ho, Wo, allow_split)
    census[(B, n_in, n_out, Ho, Wo, bool(allow_split), bool(r))] += 1
    return r


ops._wino_ok = counting_ok
real_conv2d = torch.conv2d
fallback = collections.Counter()


def counting_conv2d(x, w, b=None, stride=1, padding=0, *a, **k):
    fallback[(tuple(x.shape), tuple(w.shape), str(stride), str(padding))] += 1
    return real_conv2d(x, w, b, stride, padding, *a, **k)


torch.conv2d = counting_conv2d
torch.nn.functional.conv2d = counting_conv2d

from depthmodelhardening_amd.options import MonodepthOptions  # noqa: E402
from depthmodelhardening_amd.trainer import Trainer  # noqa: E402

argv = ["--dataset", "synthetic", "--frame_ids", "0", "--use_stereo", "--height", "320", "--width", "1024", "--batch_size",
        "32", "--learning_rate", "1e-5", "--adv_train", "--norm_type", "l_inf", "--atk_steps", "10", "--weights_init",
        "scratch", "--model_name", "census", "--log_dir", "/tmp/dmh_census", "--synthetic_len", "1000000"]
torch.backends.cudnn.benchmark = False
job = Trainer(MonodepthOptions().parse(argv), rank=0, world_size=1, device=torch.device("cuda"))
job.set_train()
job.warm_kernels()
census.clear()
fallback.clear()
job.train_step()
torch.cuda.synchronize()
print("K10 dispatch rule, one step:")
for k, n in sorted(census.items()):
    print("  B=%3d %4d->%4d out %3dx%4d split_ok=%d  -> %s  x%d" % (k[0], k[1], k[2], k[3], k[4], k[5], "K10" if k[6] else "other", n))
print("torch.conv2d calls (forward fall-through and non-3x3), one step:")
for k, n in sorted(fallback.items()):
    print("  x%s w%s stride %s pad %s  x%d" % (k + (n,)))
