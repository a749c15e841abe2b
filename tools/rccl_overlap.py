#!/usr/bin/env python3
"""What ONE GPU can show about the gradient all-reduce beside the attack -- "single-GPU, prediction".

    python3 tools/rccl_overlap.py                          # default launch geometry of K10
    DMH_K10_RESERVE_CUS=8 python3 tools/rccl_overlap.py     # K10 / K17 leave eight CUs free

Finding 1 (measured, `rocprofv3 --kernel-trace` of the r4 version of this tool): a 1-rank RCCL communicator launches NO device
kernel for an in-place all-reduce -- 13,447 kernels in the trace of three iterations, none of RCCL's.  tests/test_gpu_ddp.py
therefore proves stream / event ORDERING of the side-stream collective, not its scheduling against the convolutions.

So this tool uses a stand-in with a ring collective's launch geometry (`dmh_debug_channel_copy`: C persistent workgroups of 256
threads, no LDS, streaming 57.3 MB read + write `rounds` times -- ring all-reduce moves 2 (N-1)/N of the bucket per GPU), on the
side stream, enqueued right after backward exactly where GradBucket.start_all_reduce() enqueues the collective, while the next
iteration's attack runs on the compute stream (the overlapped order of Trainer.train_step).  Reported per configuration:
the stand-in's duration alone and beside the attack, the attack's duration alone and beside the stand-in.  A K10 workgroup owns
its CU (154 KB of LDS, 512 registers per lane): a channel workgroup can only start on a CU that is free of them, and once it is
resident, a 256-workgroup K10 launch has to wait for that CU.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from depthmodelhardening_amd import _native as N
    from depthmodelhardening_amd.options import MonodepthOptions
    from depthmodelhardening_amd.trainer import Trainer
    dev = torch.device("cuda")
    argv = ["--dataset", "synthetic", "--frame_ids", "0", "--use_stereo", "--height", "320", "--width", "1024", "--batch_size", "32",
            "--atk_batch_size", "12", "--learning_rate", "1e-5", "--adv_train", "--norm_type", "l_inf", "--atk_steps", "10",
            "--weights_init", "scratch", "--model_name", "ovl", "--log_dir", "/tmp/dmh_ovl", "--synthetic_len", "1000000"]
    tr = Trainer(MonodepthOptions().parse(argv), rank=0, world_size=1, device=dev)
    tr.set_train()
    tr.warm_kernels()
    tr.train_step()
    torch.cuda.synchronize()
    lib = N.lib()
    n = tr.bucket.numel - tr.bucket.numel % 4
    src = tr.bucket.flat[:n]
    dst = torch.empty_like(src)
    side = torch.cuda.Stream()

    def ev():
        return torch.cuda.Event(enable_timing=True)

    def attack_ms(with_copy, channels, rounds):
        a0, a1, c0, c1 = ev(), ev(), ev(), ev()
        torch.cuda.synchronize()
        if with_copy:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                c0.record(side)
                N.check(lib.dmh_debug_channel_copy(N.ptr(src), N.ptr(dst), n, channels, rounds, C.c_void_p(side.cuda_stream)))
                c1.record(side)
        a0.record()
        tr.update_adv_obj()
        a1.record()
        torch.cuda.synchronize()
        return a0.elapsed_time(a1), (c0.elapsed_time(c1) if with_copy else None)

    def copy_alone(channels, rounds):
        c0, c1 = ev(), ev()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            c0.record(side)
            N.check(lib.dmh_debug_channel_copy(N.ptr(src), N.ptr(dst), n, channels, rounds, C.c_void_p(side.cuda_stream)))
            c1.record(side)
        torch.cuda.synchronize()
        return c0.elapsed_time(c1)

    import ctypes as C
    base = sorted(attack_ms(False, 0, 0)[0] for _ in range(3))[1]
    print("K10 reserve CUs: %s; attack alone (10 steps, 12 scenes): %.2f ms" % (os.environ.get("DMH_K10_RESERVE_CUS", "0"), base))
    for channels in (8, 16, 32, 64):
        rounds = 2                  # ~ the 2 (N-1)/N bucket volumes a ring all-reduce moves per GPU
        alone = sorted(copy_alone(channels, rounds) for _ in range(3))[1]
        runs = sorted(attack_ms(True, channels, rounds) for _ in range(3))
        atk, cp = runs[1]
        print("  %2d channel workgroups, %d x 57.3 MB: stand-in alone %.3f ms; beside the attack %.3f ms; attack beside it %.2f ms "
              "(%+.2f ms)" % (channels, rounds, alone, cp, atk, atk - base))


if __name__ == "__main__":
    main()
