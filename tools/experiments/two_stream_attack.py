#!/usr/bin/env python3
"""Experiment (report only): does the attack's chain of ~3,300 short dependent launches run faster as TWO independent chains of half
the scenes on two HIP streams?  The device time of one 12-scene attack on one stream against two 6-scene attacks on two streams,
each measured with the streams held back (a spin kernel) until the host has enqueued everything, so that the host's launch rate
is not part of the number.

    python3 tools/experiments/two_stream_attack.py [--steps 10]
"""
import argparse
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from depthmodelhardening_amd import ops  # noqa: E402
from depthmodelhardening_amd.options import MonodepthOptions  # noqa: E402
from depthmodelhardening_amd.trainer import Trainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--scenes", type=int, default=12)
ap.add_argument("--hold_ms", type=float, default=400.0)
cli = ap.parse_args()
dev = torch.device("cuda:0")
argv = ["--dataset", "synthetic", "--frame_ids", "0", "--use_stereo", "--height", "320", "--width", "1024", "--batch_size", "4",
        "--learning_rate", "1e-5", "--adv_train", "--norm_type", "l_inf", "--atk_steps", str(cli.steps), "--weights_init",
        "scratch", "--model_name", "x", "--log_dir", "/tmp/dmh_two_stream", "--synthetic_len", "100000", "--atk_batch_size",
        str(cli.scenes)]
job = Trainer(MonodepthOptions().parse(argv), rank=0, world_size=1, device=dev)
job.set_train()
atk = job.dataset.depth_atk
n = cli.scenes
scenes = job.dataset.next_scenes(n)
half = [scenes[: n // 2].contiguous(), scenes[n // 2:].contiguous()]
atks = [copy.copy(atk), copy.copy(atk)]

# the spin kernel's rate
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
torch.cuda._sleep(10_000_000)
e1.record()
torch.cuda.synchronize()
cyc_per_ms = 10_000_000 / e0.elapsed_time(e1)
hold = int(cli.hold_ms * cyc_per_ms)
print("spin kernel: %.0f cycles per ms; holding the streams for %.0f ms" % (cyc_per_ms, cli.hold_ms))


def held(fn_streams):
    """fn_streams: list of (stream, callable).  All streams wait for a spin kernel on a third; returns (device ms from the end of
    the spin to the last stream's end, host ms to enqueue everything)."""
    gate_s = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(gate_s):
        torch.cuda._sleep(hold)
        gate = torch.cuda.Event(enable_timing=True)
        gate.record()
    ends = []
    t0 = time.perf_counter()
    for s, fn in fn_streams:
        s.wait_event(gate)
        with torch.cuda.stream(s):
            fn()
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ends.append(e)
    host_ms = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    return max(gate.elapsed_time(e) for e in ends), host_ms


main = torch.cuda.current_stream(dev)
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
with ops.frozen_weights():      # one scope around everything: the transformed filters are made once, on the main stream
    atk(scenes, n)
    torch.cuda.synchronize()
    for rep in range(3):
        one, h1 = held([(main, lambda: atk(scenes, n))])
        six, h6 = held([(main, lambda: atks[0](half[0], n // 2))])
        seq, hs = held([(main, lambda: (atks[0](half[0], n // 2), atks[1](half[1], n // 2)))])
        two, h2 = held([(sa, lambda: atks[0](half[0], n // 2)), (sb, lambda: atks[1](half[1], n // 2))])
        print("rep %d  %d scenes, one stream: %.2f ms (host %.0f)   %d scenes: %.2f ms (host %.0f)   %d + %d one stream: %.2f ms "
              "(host %.0f)   %d || %d two streams: %.2f ms (host %.0f)" % (rep, n, one, h1, n // 2, six, h6, n // 2, n // 2, seq, hs,
                                                                         n // 2, n // 2, two, h2), flush=True)
