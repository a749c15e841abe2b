#!/usr/bin/env python3
"""Every Winograd-convolution launch (K10 / K17) of one adversarial-training step with its shape, its work-item arithmetic and its
HIP-event time:

    python3 tools/k10_census.py [--config 2] > census.txt

per (entry point, B, C, K, H, W, pad, epilogue): launches per step, mean microseconds, work items, rounds over the 256 CUs
(items / 256), the launch's ideal time at the measured 3.05 us per 8-channel chunk with perfect balance, and what the
quantisation to whole rounds costs.  Diagnostic for DESIGN.md section 9 item 0 / VERDICT r4 item 2."""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthmodelhardening_amd import _native as N  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch_size", type=int, default=32)
ap.add_argument("--atk_batch_size", type=int, default=12)
ap.add_argument("--norm_type", default="l_inf")
cli = ap.parse_args()
lib = N.lib()
log = []
recording = [False]


def wrap(name, shape_of):
    real = getattr(lib, name)

    def f(*args):
        if not recording[0]:
            return real(*args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = real(*args)
        e1.record()
        log.append((name,) + shape_of(args) + ((e0, e1),))
        return rc
    setattr(lib, name, f)


wrap("dmh_wino_conv3x3", lambda a: (a[3], a[4], a[5], a[6], a[7], a[8], "plain"))
wrap("dmh_wino_conv3x3_act", lambda a: (a[5], a[6], a[7], a[8], a[9], a[10], "epi relu=%d res=%d" % (a[4], a[3] is not None and bool(a[3]))))
wrap("dmh_wino32_conv3x3", lambda a: (a[3], a[4], a[5], a[6], a[7], a[8], "k17"))
# the entry points with a stream-K workspace (round 5: ops.py calls these; the library decides per launch)
wrap("dmh_wino_conv3x3_ws", lambda a: (a[3], a[4], a[5], a[6], a[7], a[8], "plain"))
wrap("dmh_wino32_conv3x3_ws", lambda a: (a[3], a[4], a[5], a[6], a[7], a[8], "k17"))
wrap("dmh_wino_conv3x3_act_ws", lambda a: (a[5], a[6], a[7], a[8], a[9], a[10], "epi relu=%d res=%d" % (a[4], a[3] is not None and bool(a[3]))))

from depthmodelhardening_amd.options import MonodepthOptions  # noqa: E402
from depthmodelhardening_amd.trainer import Trainer  # noqa: E402

argv = ["--dataset", "synthetic", "--frame_ids", "0", "--use_stereo", "--height", "320", "--width", "1024", "--batch_size",
        str(cli.batch_size), "--atk_batch_size", str(cli.atk_batch_size), "--learning_rate", "1e-5", "--adv_train", "--norm_type",
        cli.norm_type, "--atk_steps", "10", "--weights_init", "scratch", "--model_name", "census", "--log_dir", "/tmp/dmh_census",
        "--synthetic_len", "1000000"]
job = Trainer(MonodepthOptions().parse(argv), rank=0, world_size=1, device=torch.device("cuda"))
job.set_train()
job.warm_kernels()
job.train_step()
torch.cuda.synchronize()
recording[0] = True
job.train_step()
torch.cuda.synchronize()
recording[0] = False
agg = collections.OrderedDict()
for name, B, C, K, H, W, pad, kind, (e0, e1) in log:
    agg.setdefault((name, B, C, K, H, W, pad, kind), []).append(e0.elapsed_time(e1) * 1e3)
rows = []
CU = 256
for (name, B, C, K, H, W, pad, kind), ts in agg.items():
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    Ht, Wt = Ho // 2, Wo // 2
    if name.startswith("dmh_wino32_conv3x3"):
        items = B * ((Wt + 31) // 32) * ((Ht + 3) // 4) * ((K + 31) // 32)
        split = 1
    else:
        narrow = (Wt % 32) != 0 and (Wt <= 16 or ((Wt + 15) // 16 * 16 - Wt) < ((Wt + 31) // 32 * 32 - Wt))
        kg = (K + 63) // 64
        if narrow:
            gx = (Wt + 15) // 16
            if 5 * Ht <= 4 * ((Ht + 3) // 4 * 4):
                regions = gx * ((B * Ht + 3) // 4) * kg
            else:
                regions = B * gx * ((Ht + 3) // 4) * kg
        else:
            regions = B * ((Wt + 31) // 32) * ((Ht + 1) // 2) * kg
        nch = C // 8
        split = 2 if (kind == "plain" and regions < 192 and nch % 2 == 0 and nch >= 6) else 1
        items = regions * split
    nch = C // 8 // split
    mean = sum(ts) / len(ts)
    rounds = items / float(CU)
    per_item = nch * 3.05 + 4.0
    ideal = rounds * per_item                       # perfect balance
    quant = -(-items // CU) * per_item              # whole rounds
    rows.append((len(ts) * mean, name.replace("dmh_", ""), B, C, K, H, W, pad, kind, len(ts), mean, items, rounds, ideal, quant))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("%-22s %3s %4s %4s %4s %5s %3s %-18s %4s %8s %6s %6s %8s %8s %9s" % (
    "entry", "B", "C", "K", "H", "W", "pad", "kind", "n", "mean us", "items", "rounds", "ideal us", "quant us", "total ms"))
for total, name, B, C, K, H, W, pad, kind, n, mean, items, rounds, ideal, quant in rows:
    print("%-22s %3d %4d %4d %4d %5d %3d %-18s %4d %8.1f %6d %6.2f %8.1f %8.1f %9.3f" % (
        name, B, C, K, H, W, pad, kind, n, mean, items, rounds, ideal, quant, total / 1e3))
print("total %.2f ms in %d launches (event time includes ~2-4 us of launch gap per launch)" % (tot / 1e3, len(log)))
