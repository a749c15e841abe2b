#!/usr/bin/env python3
"""Headline benchmark: adversarial-training images/s @1024x320, 10-step PGD-L_inf, per-GPU batch 32
(BASELINE.json ``metric``, configs[1]) on synthetic KITTI-shaped frames.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = one iteration of MD2/trainer.py:297-315: attack (Phy_obj_atk, 10 steps, 12 scenes) ->
GPU-side sample synthesis -> encoder/decoder forward -> fused photometric+SSIM+smoothness loss ->
backward -> gradient all-reduce (flat bucket, RCCL) -> Adam.  fp32 throughout, random-init ResNet-18 U-Net.
Rank 0 prints ONE JSON line with the contract fields plus ``roofline`` (dominant hand-written kernel,
HIP-event timed inside the timed region) and ``cpu_baseline`` (the CPU oracle on a bounded sample, N=1 only).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32 MFMA peak of MI355X (MI355X_MICROARCH.md)
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured float4-copy ceiling is 6290

# HBM bytes per launch measured with rocprofv3 PMC passes (2*FETCH_SIZE + WRITE_SIZE, the gfx950 correction of
# MI355X_MICROARCH.md section HBM) for the profiled shape; see profiles/README.md.  Keyed by (B, H, W).
MEASURED_TRAFFIC = {(32, 320, 1024): {"photo_fwd": 482.6e6, "photo_bwd": 641.5e6, "source": "profiles/r01_k1k2_pmc.csv"}}


def k1_bytes(B, H, W, scales=4):
    """Algorithmic HBM bytes of the fused K1 launches (all-scales-fused variant):
    target + source read once (24 B/px) + disparity pyramid; backward additionally reads the selection
    maps and writes the up-sampled disparity gradients (4 B/px/scale each)."""
    hw = H * W
    disp = sum(4 * (hw >> (2 * s)) for s in range(scales))
    fwd = B * (24 * hw + disp)
    bwd = B * (24 * hw + disp + 2 * 4 * hw * scales)
    return fwd, bwd


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


def cpu_baseline(height, width, atk_steps, batch):
    """The CPU oracle (oracle/, plain PyTorch) on a bounded sample of the same iteration, extrapolated
    linearly: attack time ~ PGD steps, train-step time ~ batch."""
    from depthmodelhardening_amd.depth_model import import_depth_model
    from oracle import train_step_ref
    torch.manual_seed(0)
    cores = usable_cores()
    torch.set_num_threads(cores)
    model = import_depth_model((1024, 320))
    s_steps, s_batch, s_ba = 1, 2, 12
    r = train_step_ref.timed_iteration(model, B_train=s_batch, Ba=s_ba, atk_steps=s_steps, H=height, W=width)
    t_iter = r["attack_s"] * (atk_steps / s_steps) + r["train_s"] * (batch / s_batch)
    return {"value": round(batch / t_iter, 4), "unit": "images/s", "cores": r["cores"], "kind": "port",
            "sample": "oracle (CPU PyTorch restatement): %d-step attack on %d scenes took %.1fs, train step at batch %d "
                      "took %.1fs; scaled linearly to %d steps / batch %d" % (s_steps, s_ba, r["attack_s"], s_batch,
                                                                           r["train_s"], atk_steps, batch)}


def main():
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
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--phases", action="store_true", help="also print a per-phase GPU-time breakdown to stderr")
    a = ap.parse_args()

    from depthmodelhardening_amd import ops
    from depthmodelhardening_amd.ddp import init_distributed
    from depthmodelhardening_amd.options import MonodepthOptions
    from depthmodelhardening_amd.trainer import Trainer

    t_start = time.perf_counter()
    rank, world, device = init_distributed("cuda")
    if world != a.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world), file=sys.stderr)
    torch.manual_seed(1234 + rank)
    # MIOpen exhaustive find (cudnn.benchmark=True) costs minutes on a fresh box with an empty perf cache, which is
    # where this benchmark always runs: stay in immediate mode unless asked
    torch.backends.cudnn.benchmark = bool(int(os.environ.get("DMH_MIOPEN_FIND", "0")))
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
    trainer = Trainer(opts, rank=rank, world_size=world, device=device)
    trainer.set_train()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    import threading
    stop_hb = threading.Event()

    def heartbeat():
        while not stop_hb.wait(60.0):
            print("[bench %.0fs] ... still running (first step: MIOpen builds its kernels; last: CPU baseline)" %
                  (time.perf_counter() - t_start), file=sys.stderr, flush=True)
    if rank == 0:
        threading.Thread(target=heartbeat, daemon=True).start()

    def note(msg):
        if rank == 0:
            print("[bench %.0fs] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    note("trainer built")
    for i in range(a.warmup):
        trainer.train_step()
        torch.cuda.synchronize()
        note("warmup step %d done" % i)
    trainer._apply_pending_update()
    sync()
    ops.enable_profile(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses = trainer.train_step()
    trainer._apply_pending_update()
    sync()
    elapsed = time.perf_counter() - t0
    note("timed region done: %.3fs for %d steps" % (elapsed, a.steps))
    kms = {k: v for k, v in ops.profile_ms().items() if k.startswith("photo_")}
    kbytes = ops.profile_bytes()
    ops.enable_profile(False)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(losses["loss"].detach())

    if a.phases and rank == 0:
        phase_breakdown(trainer)

    if rank == 0:
        fwd_b, bwd_b = k1_bytes(a.batch_size, a.height, a.width)
        names = {"photo_fwd": ("photo_fwd_kernel", fwd_b), "photo_bwd": ("photo_bwd_kernel", bwd_b)}
        dom = max(kms, key=lambda k: kms[k]) if kms else None
        roof = None
        if dom:
            kname, nbytes = names[dom]
            achieved = nbytes / (kms[dom] * 1e-3) / 1e9
            meas = MEASURED_TRAFFIC.get((a.batch_size, a.height, a.width), {})
            roof = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": meas.get(dom),
                    "traffic_source": meas.get("source"),
                    "note": "all-scales-fused kernel: traffic == algorithmic bytes (no re-reads); it is bound by VALU issue "
                            "and dependent-load latency, not by HBM (profiles/README.md)",
                    "avg_ms": round(kms[dom], 4), "algorithmic_bytes": nbytes,
                    "others": {names[k][0]: {"avg_ms": round(v, 4), "GB/s": round(names[k][1] / (v * 1e-3) / 1e9, 1)}
                               for k, v in kms.items() if k != dom}}
            # streaming kernels of the decoder glue: shapes vary per launch, so total bytes / total time
            for k, (cnt, ms, nb, fl) in sorted(kbytes.items()):
                if not k.startswith("photo_") and ms > 0:
                    ent = {"launches_per_step": round(cnt / a.steps, 1), "ms_per_step": round(ms / a.steps, 3)}
                    if fl > 0:      # the Winograd-MFMA convolution (K10): bound by the fp32 matrix pipe, not by HBM
                        direct = fl / (ms * 1e-3) / 1e12
                        ent.update({"bound": "mfma", "TFLOP/s_direct_equivalent": round(direct, 1),
                                    "achieved": round(direct / 2.25, 1), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": round(direct / 2.25 / MFMA_F32_PEAK_TFLOPS, 4),
                                    "note": "achieved = MFMA flops actually issued (Winograd F(2x2,3x3): direct / 2.25)"})
                    else:
                        ent.update({"GB/s": round(nb / (ms * 1e-3) / 1e9, 1),
                                    "frac": round(nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)})
                    roof["others"][k + "_kernel"] = ent
        out = {"metric": "adv-train images/sec @1024x320, 10-step PGD, bs32", "value": round(a.batch_size * world * a.steps / elapsed, 3),
               "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "Monodepth2 ResNet18 %dx%d, %d-step %s attack on 12 scenes, train batch %d/GPU, "
                                      "stereo photometric+SSIM+smoothness loss (%s)%s%s, Adam" % (
                                          a.width, a.height, a.atk_steps,
                                          "PGD-L_inf" if a.norm_type == "l_inf" else "L0/Adam", a.batch_size, a.loss_variant,
                                          " + supervised_adv" if a.supervised_adv else "",
                                          " + contrastive" if a.contrastive_learning else ""),
                          "global_batch": a.batch_size * world, "parallelism": "dp%d" % world,
                          "attack_overlap": bool(world > 1 and not a.sync_attack), "final_loss": round(loss_val, 6)},
               "roofline": roof}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.height, a.width, a.atk_steps, a.batch_size)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def phase_breakdown(trainer, iters=3):
    """GPU time of attack / synthesis / forward+loss / backward / optimiser, by CUDA events (diagnostic)."""
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
