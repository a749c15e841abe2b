#!/usr/bin/env python3
"""Which ATen operators still launch kernels inside one adversarial-training step, with input shapes, call counts and device
time (torch.profiler, grouped by input shape) -- the residue VERDICT r4 item 5 asks to retire:

    python3 tools/aten_census.py [--top 40] [--stacks]
"""
import argparse
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthmodelhardening_amd.options import MonodepthOptions  # noqa: E402
from depthmodelhardening_amd.trainer import Trainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--top", type=int, default=45)
ap.add_argument("--stacks", action="store_true")
ap.add_argument("--norm_type", default="l_inf")
cli = ap.parse_args()
argv = ["--dataset", "synthetic", "--frame_ids", "0", "--use_stereo", "--height", "320", "--width", "1024", "--batch_size", "32",
        "--atk_batch_size", "12", "--learning_rate", "1e-5", "--adv_train", "--norm_type", cli.norm_type, "--atk_steps", "10",
        "--weights_init", "scratch", "--model_name", "census", "--log_dir", "/tmp/dmh_census", "--synthetic_len", "1000000"]
job = Trainer(MonodepthOptions().parse(argv), rank=0, world_size=1, device=torch.device("cuda"))
job.set_train()
job.warm_kernels()
job.train_step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=cli.stacks) as prof:
    job.train_step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6 if cli.stacks else 0):
    dev = getattr(e, "self_device_time_total", None)
    if dev is None:
        dev = getattr(e, "self_cuda_time_total", 0)
    if dev > 0 and e.key.startswith("aten::"):
        rows.append((dev, e.key, e.count, str(e.input_shapes)[:150], e.stack if cli.stacks else None))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("ATen operators with device time in ONE step: %.3f ms in total" % (tot / 1e3))
for dev, key, cnt, shapes, stack in rows[:cli.top]:
    print("%9.1f us %5d x  %-28s %s" % (dev, cnt, key, shapes))
    if stack:
        for ln in stack[:6]:
            if "depthmodelhardening_amd" in ln or "bench" in ln:
                print("                      %s" % ln.strip()[:160])
