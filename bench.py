#!/usr/bin/env python3
"""Headline benchmark: adversarial-training images/s @1024x320, 10-step PGD-L_inf, per-GPU batch 32
(BASELINE.json ``metric``, configs[1]) on synthetic KITTI-shaped frames.

    python bench.py --gpus N --steps K --warmup W            # starts the N ranks itself (one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = one iteration of MD2/trainer.py:297-315: attack (Phy_obj_atk, 10 steps, 12 scenes) ->
GPU-side sample synthesis -> encoder/decoder forward -> fused photometric+SSIM+smoothness loss ->
backward -> gradient all-reduce (flat bucket, RCCL) -> Adam.  fp32 throughout, random-init ResNet-18 U-Net.
Rank 0 prints ONE JSON line with the contract fields plus ``roofline`` (dominant K1 kernel, HIP-event timed inside
the timed region) and ``cpu_baseline`` (the CPU oracle on bounded samples, N=1 only).

Launcher: with ``--gpus N`` (N > 1) and no torchrun environment, this process starts N children (RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set) BEFORE anything touches the GPU, relays rank 0's
JSON line and exits with the worst child's code.  It never re-execs a process that has initialised the GPU.
"""
import argparse
import json
import os
import platform
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32 MFMA peak of MI355X (MI355X_MICROARCH.md)
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured float4-copy ceiling is 6290

# HBM bytes per launch measured with rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE, separate passes) for the profiled
# shape, keyed by (B, H, W); see profiles/README.md.  FETCH_SIZE is taken 1:1: these kernels read one dword per lane,
# and the x2 correction of MI355X_MICROARCH.md section HBM is for 16-byte-per-lane streaming reads only ("other access
# widths are uncalibrated: calibrate on a known byte count in your own access pattern") -- calibrated on
# smooth_fwd_kernel, which reads its 222.9 MB exactly once and reports FETCH_SIZE = 198.8 MB.  The backward reports
# fewer bytes than its algorithmic reads because the images are still in the 256 MB Infinity Cache from the forward.
MEASURED_TRAFFIC = {(32, 320, 1024): {"photo_fwd": 328.9e6, "photo_bwd": 218.9e6, "source": "profiles/r02_k1k2_pmc.csv"}}


def k1_bytes(B, H, W, scales=4):
    """Algorithmic HBM bytes of the fused K1 launches, SURVEY.md section 8d, all-scales-fused variant: forward reads
    target + source once (24 B/px) + the disparity pyramid (4*sum hw_s) and writes scalars only; backward
    additionally writes the disparity gradients (4*sum hw_s).  The kernel's own intermediates (selection byte,
    staging blocks) are NOT algorithmic: they show up in traffic / algorithmic."""
    hw = H * W
    disp = sum(4 * (hw >> (2 * s)) for s in range(scales))
    return B * (24 * hw + disp), B * (24 * hw + 2 * disp)


def usable_cores():
    """Cores this process may actually use: affinity mask and cgroup CPU quota, not the host's core count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    # a GPU box hands each GPU a 16-core share even where nproc reports the whole host (256): more threads than
    # that only oversubscribe (measured: 400 s instead of 20 s for the same sample)
    return max(1, min(n, int(os.environ.get("DMH_CPU_THREADS", "16"))))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return platform.processor() or "unknown"


def cpu_baseline(height, width, atk_steps, batch, gpu_loss_ms=None):
    """SURVEY.md section 8d "CPU baseline beside it": the CPU oracle (oracle/, plain PyTorch restatement of the
    reference iteration) on the host cores, outside the timed region:
      (i)   BASELINE config 1 in full: B=2, 192x640, 3-step attack on 2 scenes, median of 3 iterations;
      (ii)  config 2 with the full 10-step attack on 12 scenes and a train step at batch 4, the train step scaled x8
            to batch 32 (the only extrapolation) -> ``value``;
      (iii) the loss path alone (generate_images_pred + compute_losses + backward) at the full B=32 shape -- the
            like-for-like of K1 + K2."""
    import torch
    from depthmodelhardening_amd.depth_model import import_depth_model
    from oracle import train_step_ref
    torch.manual_seed(0)
    cores = usable_cores()
    torch.set_num_threads(cores)
    model = import_depth_model((1024, 320))
    # (i) config 1 (the reference's CPU-runnable plumbing case)
    c1 = []
    for _ in range(3):
        r = train_step_ref.timed_iteration(model, B_train=2, Ba=2, atk_steps=3, H=192, W=640)
        c1.append(r["attack_s"] + r["train_s"])
    c1.sort()
    # (ii) config 2: full attack, train batch 4 (x8)
    s_batch = 4
    r = train_step_ref.timed_iteration(model, B_train=s_batch, Ba=12, atk_steps=atk_steps, H=height, W=width)
    t_iter = r["attack_s"] + r["train_s"] * (batch / s_batch)
    # (iii) loss path at the full shape
    t_loss = train_step_ref.timed_loss_path(batch, height, width)
    out = {"value": round(batch / t_iter, 4), "unit": "images/s", "cores": r["cores"], "kind": "port",
           "cpu_model": cpu_model(),
           "sample": "oracle (CPU PyTorch restatement), %d threads: %d-step attack on 12 scenes %.1fs + train step at batch %d "
                     "%.1fs scaled x%d to batch %d (only extrapolation)" % (r["cores"], atk_steps, r["attack_s"], s_batch,
                                                                           r["train_s"], batch // s_batch, batch),
           "config1_full": {"images_per_s": round(2 / c1[1], 4), "s_per_iteration_median_of_3": round(c1[1], 3),
                            "what": "B=2, 192x640, 3-step PGD on 2 scenes, attack + train step, no extrapolation"},
           "loss_path_full_shape": {"cpu_s": round(t_loss, 3), "gpu_ms": gpu_loss_ms,
                                    "what": "generate_images_pred + compute_losses + backward at B=%d %dx%d "
                                            "(CPU oracle) vs K1+K2 forward+backward (HIP events)" % (batch, width, height)}}
    if gpu_loss_ms:
        out["loss_path_full_shape"]["speedup"] = round(t_loss * 1e3 / gpu_loss_ms, 1)
    return out


# ---------------------------------------------------------------------------------------------------------------
# launcher (no GPU call may happen before this returns in the parent)
# ---------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv, device_type="cuda"):
    """Start ``n`` rank processes of this script (one per GPU), wait, relay rank 0's stdout.  Returns the exit code."""
    port = int(os.environ.get("MASTER_PORT") or _free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DMH_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = subprocess.PIPE if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out))
    line = procs[0].communicate()[0].decode() if procs else ""
    rc = 0
    for p in procs:
        p.wait()
        rc = max(rc, abs(p.returncode))
    sys.stdout.write(line)
    sys.stdout.flush()
    return rc


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch_size", type=int, default=32)
    ap.add_argument("--height", type=int, default=320)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--atk_steps", type=int, default=10)
    ap.add_argument("--norm_type", type=str, default="l_inf", choices=["l_inf", "l_0"])
    ap.add_argument("--sync_attack", action="store_true")
    # the other BASELINE.json configs (parity/regression cases, not the headline line)
    ap.add_argument("--supervised_adv", action="store_true")
    ap.add_argument("--contrastive_learning", action="store_true")
    ap.add_argument("--loss_variant", type=str, default="md2", choices=["md2", "dh"])
    ap.add_argument("--harness", type=str, default="trainer", choices=["trainer", "physical"],
                    help="physical = BASELINE config 5: the physical_adv_training.py hardening loop (patch attack)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--device", type=str, default="cuda", choices=["cuda", "cpu"],
                    help="cpu: launcher / collective plumbing test only (gloo); no kernels run")
    ap.add_argument("--phases", action="store_true", help="also print a per-phase GPU-time breakdown to stderr")
    return ap.parse_args(argv)


def main():
    a = parse()
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world_env == 1 and not os.environ.get("DMH_BENCH_CHILD"):
        sys.exit(launch_ranks(a.gpus, sys.argv[1:], a.device))      # parent: never initialises the GPU
    if a.device == "cpu":
        return plumbing_only(a)
    run_rank(a)


def plumbing_only(a):
    """``--device cpu``: exercises the launcher, the rendezvous, the barriers and the max-over-ranks reduction with
    gloo and no kernels (tests/test_ddp_gloo.py).  Prints a JSON line with value null."""
    import torch
    import torch.distributed as dist
    from depthmodelhardening_amd.ddp import init_distributed
    rank, world, _ = init_distributed("cpu")
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the job has %d ranks" % (a.gpus, world))
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert dist.get_world_size() == a.gpus
    if rank == 0:
        print(json.dumps({"metric": "launcher plumbing (no kernels)", "value": None, "n_gpus": world,
                          "max_over_ranks": float(t.item())}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_rank(a):
    import torch
    import torch.distributed as dist
    from depthmodelhardening_amd import ops
    from depthmodelhardening_amd.ddp import init_distributed
    from depthmodelhardening_amd.options import MonodepthOptions
    from depthmodelhardening_amd.trainer import Trainer

    t_start = time.perf_counter()
    rank, world, device = init_distributed("cuda")
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the job has %d ranks (start it as `python bench.py --gpus %d` or "
                         "under torch.distributed.run with --nproc-per-node %d)" % (a.gpus, world, a.gpus, a.gpus))
    if world > 1:
        assert dist.get_world_size() == a.gpus and dist.get_backend() in ("nccl", os.environ.get("DMH_DIST_BACKEND", "nccl"))
    torch.manual_seed(1234 + rank)
    # MIOpen exhaustive find (cudnn.benchmark=True) costs minutes on a fresh box with an empty perf cache, which is
    # where this benchmark always runs: stay in immediate mode unless asked
    torch.backends.cudnn.benchmark = bool(int(os.environ.get("DMH_MIOPEN_FIND", "0")))

    def note(msg):
        if rank == 0:
            print("[bench %.0fs] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    import threading
    stop_hb = threading.Event()

    def heartbeat():
        while not stop_hb.wait(60.0):
            print("[bench %.0fs] ... still running (first step: MIOpen builds its kernels; last: CPU baseline)" %
                  (time.perf_counter() - t_start), file=sys.stderr, flush=True)
    if rank == 0:
        threading.Thread(target=heartbeat, daemon=True).start()

    if a.harness == "physical":
        from depthmodelhardening_amd import physical_adv_training as pat
        job = pat.BenchJob(a.batch_size, a.atk_steps, rank, world, device)
        workload = ("physical_adv_training.py hardening loop: Phy_obj_atk (EOT patch attack, %d steps) on %d scenes/GPU "
                    "-> frozen model disparity -> MSE -> Adam, Monodepth2 ResNet18 %dx%d" % (
                        a.atk_steps, a.batch_size, a.width, a.height))
    else:
        argv = ["--dataset", "synthetic", "--frame_ids", "0", "--use_stereo", "--height", str(a.height), "--width",
                str(a.width), "--batch_size", str(a.batch_size), "--learning_rate", "1e-5", "--adv_train", "--norm_type",
                a.norm_type, "--atk_steps", str(a.atk_steps), "--weights_init", "scratch", "--model_name", "bench",
                "--log_dir", os.path.join("/tmp", "dmh_bench_%d" % rank), "--synthetic_len", "1000000"]
        if a.sync_attack:
            argv.append("--sync_attack")
        if a.supervised_adv:
            argv.append("--supervised_adv")
        if a.contrastive_learning:
            argv.append("--contrastive_learning")
        argv += ["--loss_variant", a.loss_variant]
        opts = MonodepthOptions().parse(argv)
        job = Trainer(opts, rank=rank, world_size=world, device=device)
        job.set_train()
        workload = ("Monodepth2 ResNet18 %dx%d, %d-step %s attack on 12 scenes, train batch %d/GPU, stereo "
                    "photometric+SSIM+smoothness loss (%s)%s%s, Adam" % (
                        a.width, a.height, a.atk_steps, "PGD-L_inf" if a.norm_type == "l_inf" else "L0/Adam",
                        a.batch_size, a.loss_variant, " + supervised_adv" if a.supervised_adv else "",
                        " + contrastive" if a.contrastive_learning else ""))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    note("job built")
    # The first iteration makes MIOpen compile its kernels (minutes on a fresh box) into the per-user kernel cache.
    # With several ranks, rank 0 does that alone first and the others then find the kernels in the cache: N
    # concurrent builds of the same kernels would multiply the CPU work and contend on the cache's database lock.
    if world > 1:
        if rank == 0:
            job.warm_kernels()
            torch.cuda.synchronize()
            note("rank 0 warmed the kernel cache")
        dist.barrier()
        if rank != 0:
            job.warm_kernels()
        sync()
    for i in range(a.warmup):
        job.train_step()
        torch.cuda.synchronize()
        note("warmup step %d done" % i)
    job._apply_pending_update()
    sync()
    # Inside the timed region only the K1 launches (the kernels of the roofline entry) carry HIP events: an event pair costs
    # ~10 us of dispatch latency around a launch, and a step has ~1,300 instrumented launches (timing all of them cost
    # 8.5 ms of a 174 ms step).  The other kernels' durations (roofline.others) come from ONE extra, untimed, fully
    # instrumented step BEFORE the timed region (so that the timed region is the tail of a kernel trace).
    ops.enable_profile(True)
    job.train_step()
    job._apply_pending_update()
    sync()
    kbytes = ops.profile_bytes()
    ops.enable_profile(True, only=("photo_",))
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses = job.train_step()
    job._apply_pending_update()
    sync()
    elapsed = time.perf_counter() - t0
    note("timed region done: %.3fs for %d steps" % (elapsed, a.steps))
    kms = {k: v for k, v in ops.profile_ms().items() if k.startswith("photo_")}
    ops.enable_profile(False)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(losses["loss"].detach())

    if a.phases and rank == 0 and a.harness == "trainer":
        phase_breakdown(job)

    if rank == 0:
        fwd_b, bwd_b = k1_bytes(a.batch_size, a.height, a.width)
        names = {"photo_fwd": ("photo_fwd_kernel", fwd_b), "photo_bwd": ("photo_bwd_kernel", bwd_b)}
        dom = max(kms, key=lambda k: kms[k]) if kms else None
        roof = None
        if dom:
            kname, nbytes = names[dom]
            achieved = nbytes / (kms[dom] * 1e-3) / 1e9
            meas = MEASURED_TRAFFIC.get((a.batch_size, a.height, a.width), {})
            traffic = meas.get(dom)
            roof = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                    "traffic_over_algorithmic": round(traffic / nbytes, 3) if traffic else None,
                    "traffic_source": meas.get("source"),
                    "note": "algorithmic bytes = SURVEY 8d fused-variant figure; the fused loss does ~1,000 VALU instructions "
                            "per pixel on 30 B of traffic, so it sits on the VALU-issue roof, far below the HBM roof "
                            "(profiles/README.md)",
                    "avg_ms": round(kms[dom], 4), "algorithmic_bytes": nbytes,
                    "others": {names[k][0]: {"avg_ms": round(v, 4), "GB/s": round(names[k][1] / (v * 1e-3) / 1e9, 1),
                                             "algorithmic_bytes": names[k][1], "traffic": meas.get(k)}
                               for k, v in kms.items() if k != dom}}
            # streaming kernels of the decoder glue: shapes vary per launch, so total bytes / total time
            for k, (cnt, ms, nb, fl) in sorted(kbytes.items()):
                if not k.startswith("photo_") and ms > 0:
                    ent = {"launches_per_step": cnt, "ms_per_step": round(ms, 3)}      # the one instrumented step
                    if fl > 0 and k.startswith("wino_"):    # the Winograd-MFMA convolution (K10): bound by the fp32 matrix pipe
                        direct = fl / (ms * 1e-3) / 1e12
                        ent.update({"bound": "mfma", "TFLOP/s_direct_equivalent": round(direct, 1),
                                    "achieved": round(direct / 2.25, 1), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": round(direct / 2.25 / MFMA_F32_PEAK_TFLOPS, 4),
                                    "note": "achieved = MFMA flops actually issued (Winograd F(2x2,3x3): direct / 2.25)"})
                    elif fl > 0:                            # direct MFMA convolution (K14 stem): flops as counted
                        tf = fl / (ms * 1e-3) / 1e12
                        ent.update({"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4)})
                    else:
                        ent.update({"GB/s": round(nb / (ms * 1e-3) / 1e9, 1),
                                    "frac": round(nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)})
                    roof["others"][k + "_kernel"] = ent
        out = {"metric": "adv-train images/sec @1024x320, 10-step PGD, bs32", "value": round(a.batch_size * world * a.steps / elapsed, 3),
               "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": workload, "global_batch": a.batch_size * world, "parallelism": "dp%d" % world,
                          "attack_overlap": bool(world > 1 and not a.sync_attack and a.harness == "trainer"),
                          "final_loss": round(loss_val, 6)},
               "roofline": roof}
        if world == 1 and not a.no_cpu_baseline:
            gpu_loss_ms = round(sum(kms.values()), 4) if kms else None
            out["cpu_baseline"] = cpu_baseline(a.height, a.width, a.atk_steps, a.batch_size, gpu_loss_ms)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def phase_breakdown(trainer, iters=3):
    """GPU time of attack / synthesis / forward+loss / backward / optimiser, by CUDA events (diagnostic)."""
    import torch
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    acc = {}
    for _ in range(iters):
        marks = [ev() for _ in range(6)]
        marks[0].record()
        trainer.dataset.update_adv_obj(trainer.dataset.next_scenes(trainer.adv_args["batch_size"]))
        marks[1].record()
        inputs = trainer.dataset.next_batch(trainer.opt.batch_size)
        marks[2].record()
        outputs, losses = trainer.process_batch(inputs)
        marks[3].record()
        trainer.bucket.zero()
        losses["loss"].backward()
        marks[4].record()
        trainer.model_optimizer.step()
        marks[5].record()
        torch.cuda.synchronize()
        for i, n in enumerate(["attack", "synthesis", "forward+loss", "backward", "adam"]):
            acc[n] = acc.get(n, 0.0) + marks[i].elapsed_time(marks[i + 1]) / iters
    print("phase ms: " + json.dumps({k: round(v, 2) for k, v in acc.items()}), file=sys.stderr, flush=True)


if __name__ == "__main__":
    main()
