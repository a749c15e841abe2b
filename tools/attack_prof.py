#!/usr/bin/env python3
"""The attack phase of a config-2 step by itself (10 PGD steps on 12 scenes, windows on), for a kernel trace:

    rocprofv3 --kernel-trace --stats -d gpurun_out/atk -- python3 tools/attack_prof.py [iters]

prints, without the profiler, the phase's wall time by CUDA events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthmodelhardening_amd.options import MonodepthOptions  # noqa: E402
from depthmodelhardening_amd.trainer import Trainer  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
argv = ["--dataset", "synthetic", "--frame_ids", "0", "--use_stereo", "--height", "320", "--width", "1024", "--batch_size", "32",
        "--learning_rate", "1e-5", "--adv_train", "--norm_type", "l_inf", "--atk_steps", "10", "--weights_init", "scratch",
        "--model_name", "atk", "--log_dir", "/tmp/dmh_atk", "--synthetic_len", "1000000"]
job = Trainer(MonodepthOptions().parse(argv), rank=0, world_size=1, device=torch.device("cuda"))
job.set_train()
job.warm_kernels()
for _ in range(2):
    job.dataset.update_adv_obj(job.dataset.next_scenes(job.adv_args["batch_size"]))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
scenes = [job.dataset.next_scenes(job.adv_args["batch_size"]) for _ in range(iters)]
torch.cuda.synchronize()
e0.record()
for s in scenes:
    job.dataset.update_adv_obj(s)
e1.record()
torch.cuda.synchronize()
print("attack phase: %.2f ms per attack (10 steps, 12 scenes)" % (e0.elapsed_time(e1) / iters))
