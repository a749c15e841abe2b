#!/usr/bin/env python3
"""What can be shown of the gradient all-reduce's overlap on ONE GPU ("single-GPU, prediction").

    export TMPDIR=/tmp; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/rccl_ovl -- python3 tools/rccl_overlap.py
    [DMH_K10_RESERVE_CUS=8 exported before rocprofv3 for the second run]
    python3 tools/rccl_overlap.py --report gpurun_out/rccl_ovl          # after the run: parse the kernel trace

The process initialises backend nccl (= RCCL) with world_size 1, builds the bench's trainer (config 2: 1024x320, 12 attack
scenes, batch 32) and runs three iterations in the overlapped order of Trainer.train_step with the collective FORCED although
there is one rank: the 57.3 MB flat bucket is handed to RCCL on the side stream right after backward, the next iteration's
attack is enqueued on the compute stream, then the optimiser waits for the collective's event.  With one rank RCCL moves no
bytes over xGMI, but its device kernel is launched, scheduled against the persistent K10 workgroups and timed: the trace says
WHERE it runs -- beside K10 launches, or only in the gaps between them -- and when it finishes relative to the attack.
"""
import argparse
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29541"), RANK="0", WORLD_SIZE="1",
                      LOCAL_RANK="0", DMH_DIST_FORCE_INIT="1")
    os.environ.pop("DMH_DIST_BACKEND", None)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from depthmodelhardening_amd.ddp import GradBucket, init_distributed
    from depthmodelhardening_amd.options import MonodepthOptions
    from depthmodelhardening_amd.trainer import Trainer
    r, w, dev = init_distributed("cuda")
    assert dist.get_backend() == "nccl" and w == 1
    argv = ["--dataset", "synthetic", "--frame_ids", "0", "--use_stereo", "--height", "320", "--width", "1024", "--batch_size", "32",
            "--atk_batch_size", "12", "--learning_rate", "1e-5", "--adv_train", "--norm_type", "l_inf", "--atk_steps", "10",
            "--weights_init", "scratch", "--model_name", "ovl", "--log_dir", "/tmp/dmh_ovl", "--synthetic_len", "1000000"]
    tr = Trainer(MonodepthOptions().parse(argv), rank=0, world_size=1, device=dev)
    fc = {id(p) for p in tr.models["encoder"].encoder.fc.parameters()}
    tr.bucket = GradBucket([p for p in tr.parameters_to_train if id(p) not in fc], world_size=1, force_collective=True)
    tr.set_train()
    tr.warm_kernels()
    tr.train_step()
    torch.cuda.synchronize()
    marks = []
    for it in range(3):
        # the overlapped order of Trainer.train_step for world_size > 1: attack first (it reads weights one step old), then
        # wait for the previous iteration's collective and apply the update
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        tr.update_adv_obj()
        e1.record()
        tr._apply_pending_update()
        inputs = tr.dataset.next_batch(tr.opt.batch_size)
        _, losses = tr.process_batch(inputs)
        with tr.bucket.released():
            losses["loss"].backward()
        tr.bucket.start_all_reduce()          # side stream; the NEXT iteration's attack is enqueued before anybody waits on it
        tr._pending = True
        e2.record()
        marks.append((e0, e1, e2))
    tr._apply_pending_update()
    torch.cuda.synchronize()
    for i, (e0, e1, e2) in enumerate(marks):
        print("iteration %d: attack %.2f ms, forward/backward %.2f ms" % (i, e0.elapsed_time(e1), e1.elapsed_time(e2)))
    dist.barrier()
    dist.destroy_process_group()


def report(folder):
    files = glob.glob(os.path.join(folder, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        print("no kernel trace under", folder)
        return 1
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    nccl = [r for r in rows if "nccl" in r[2].lower() or "rccl" in r[2].lower()]
    k10 = [r for r in rows if "wino_conv_kernel" in r[2] or "wino32_conv_kernel" in r[2] or "wino_wrw_kernel" in r[2]]
    print("%d kernels in the trace, %d RCCL kernels, %d persistent MFMA-convolution launches (K10 / K17 / K18)" % (len(rows), len(nccl), len(k10)))
    for s, e, name in nccl:
        inside = [(a, b, n) for a, b, n in k10 if a < e and b > s]
        ovl = sum(min(e, b) - max(s, a) for a, b, n in inside)
        others = [(a, b, n) for a, b, n in rows if a < e and b > s and "nccl" not in n.lower()]
        # which kernel was running when it started / how long after the previous compute kernel's end it started
        running = [n for a, b, n in rows if a <= s < b and "nccl" not in n.lower()]
        print("  %-60s %8.1f us | overlaps %d K10-class launches for %.1f us (%.0f %% of its duration), %d kernels of the compute "
              "stream ran meanwhile | started while running: %s" % (name[:60], (e - s) / 1e3, len(inside), ovl / 1e3, 100.0 * ovl / max(1, e - s),
                                                                   len(others), (running[0][:50] if running else "nothing (a gap)")))
    return 0


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--report", default=None)
    a = ap.parse_args()
    sys.exit(report(a.report) if a.report else run())
