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

BASELINE_CONFIGS = {      # BASELINE.json "configs" (index = position in that list), and the flags each one sets here
    2: ("Monodepth2 ResNet18, 1024x320, 10-step PGD-L_inf, bs32, 1xMI355X", {}),
    3: ("Monodepth2 --adv_train --norm_type l_0 + supervised_adv, 1024x320, bs32, 8xMI355X DDP",
        {"norm_type": "l_0", "supervised_adv": True}),
    4: ("DepthHints DH_MS_320_1024, 20-step PGD + contrastive_learning, bs64, 8xMI355X",
        {"batch_size": 64, "atk_steps": 20, "loss_variant": "dh", "contrastive_learning": True}),
    5: ("physical_adv_training.py patch-attack (physicalTrans EOT) on Monodepth2, 1024x320, bs32, 8xMI355X",
        {"harness": "physical"}),
}


K1_SOURCES = ("depthmodelhardening_amd/csrc/photo_loss.hip", "depthmodelhardening_amd/csrc/smooth_loss.hip")


def k1_source_hash():
    """sha256 over the sources the K1 / K2 kernels are compiled from: the key that ties a committed counter summary to the
    code it was measured on (tools/summarize_profiles.py writes it into profiles/rNN_k1k2_shape.json)."""
    import hashlib
    h = hashlib.sha256()
    for rel in K1_SOURCES:
        with open(os.path.join(REPO, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def measured_traffic(B, H, W):
    """HBM bytes per launch of the K1 kernels from the newest committed rocprofv3 PMC summary (profiles/rNN_k1k2_pmc.csv:
    FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes by tools/collect_profiles.sh, unit KiB), valid for the
    shape that summary was taken at AND for the kernel source it was taken on (profiles/rNN_k1k2_shape.json: B, H, W and
    ``k1_source_sha256``): counters of another shape or of an older photo_loss.hip are NOT reported -- ``stale`` says why and
    the JSON's ``traffic`` is null.  FETCH_SIZE is taken 1:1: these kernels read one
    dword per lane, and the x2 correction of MI355X_MICROARCH.md section HBM is for 16-byte-per-lane streaming reads
    only ("other access widths are uncalibrated: calibrate on a known byte count in your own access pattern") --
    calibrated on smooth_fwd_kernel, which reads its 222.9 MB exactly once and reports FETCH_SIZE = 198.8 MB.  The
    backward reports fewer bytes than its algorithmic reads when the images are still in the 256 MB Infinity Cache from
    the forward."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_k1k2_pmc.csv")))
    if not files:
        return {"stale": "no profiles/rNN_k1k2_pmc.csv"}
    f = files[-1]
    shape_file = f.replace("_pmc.csv", "_shape.json")
    shape = json.load(open(shape_file)) if os.path.exists(shape_file) else {}
    src = os.path.relpath(f, REPO)
    if (shape.get("B"), shape.get("H"), shape.get("W")) != (B, H, W):
        return {"stale": "%s was collected at another shape" % src}
    if shape.get("k1_source_sha256") != k1_source_hash():
        return {"stale": "%s was collected on another version of the kernel source (%s != %s): re-run tools/collect_profiles.sh"
                         % (src, shape.get("k1_source_sha256"), k1_source_hash())}
    out = {"source": src}
    for r in csv.DictReader(open(f)):
        for key in ("photo_fwd", "photo_bwd"):
            if key + "_kernel" in r["Kernel"]:
                out[key] = (float(r["FETCH_SIZE"]) + float(r["WRITE_SIZE"])) * 1024.0
                if r.get("SQ_INSTS_VALU"):
                    out[key + "_valu_insts"] = float(r["SQ_INSTS_VALU"])
                if r.get("GRBM_GUI_ACTIVE"):
                    out[key + "_gui_active"] = float(r["GRBM_GUI_ACTIVE"])
    return out


# A wave64 vector instruction that is not packed occupies its SIMD's issue port for 4 cycles (MI355X_MICROARCH.md, constants
# table: "vector-instruction ISSUE cost ... v_add_f32 / v_fma_f32 4"; the 2-cycle figure of the data sheet is reached by packed
# v_pk_*_f32 only: tools/micro/valu_rate.hip measured 4.44 nominal-clock cycles for scalar fp32 FMAs at 8 waves per SIMD and
# 4.83 for packed ones, profiles/r04_valu_rate.txt).  The roof is counted in MEASURED cycles: GRBM_GUI_ACTIVE of the same counter
# run (8 XCDs) is the kernel's duration in shader cycles at the clock the chip really held (2.2 GHz under this load, not 2.4).
VALU_ISSUE_CYCLES = 4.0
N_SIMD, N_XCD = 1024, 8
VALU_F32_SCALAR_TFLOPS = round(N_SIMD * 64 * 2 * 2.4e9 / 4.44 / 1e12, 1)     # 70.9: K12's roof (measured saturated issue rate)


def valu_roof(insts, measured_ms, gui_active=None):
    """The vector-issue roof of a VALU-bound kernel: SQ_INSTS_VALU wave-instructions x 4 issue cycles on 1024 SIMDs, against
    the kernel's duration in shader cycles (GRBM_GUI_ACTIVE / 8 XCDs, same counter collection).  ``frac`` = the share of the
    kernel's cycles in which its SIMDs' vector issue ports are taken; by construction <= 1.  Without the cycle counter the
    nominal 2.4 GHz converts the HIP-event time instead, and the entry says so."""
    per_simd = insts / float(N_SIMD)
    out = {"bound": "valu", "SQ_INSTS_VALU": insts, "issue_cycles_per_wave_instruction": VALU_ISSUE_CYCLES}
    if gui_active:
        cyc = gui_active / N_XCD
        out.update({"kernel_cycles": round(cyc), "clock": "measured (GRBM_GUI_ACTIVE / 8)",
                    "measured_cycles_per_wave_instruction_per_simd": round(cyc / per_simd, 3),
                    "frac": round(VALU_ISSUE_CYCLES * per_simd / cyc, 4)})
        if measured_ms:
            out["clock_ghz_during_kernel"] = round(cyc / (measured_ms * 1e-3) / 1e9, 3)
    elif measured_ms:
        cyc = measured_ms * 1e-3 * 2.4e9
        out.update({"clock": "nominal 2.4 GHz (no cycle counter in the summary): an upper bound on the cycles, so a lower "
                             "bound on frac", "frac": round(min(1.0, VALU_ISSUE_CYCLES * per_simd / cyc), 4)})
    out["data_sheet_2_cycle_floor_ms"] = round(insts * 2.0 / N_SIMD / 2.4e9 * 1e3, 4)
    return out


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


def unet_flops(height, width):
    """FLOPs of ONE image through the ResNet-18 U-Net at this resolution, measured with torch.utils.flop_counter on the
    reference module path (meta tensors: nothing is computed): forward, backward to the input only (what an attack
    step runs: the weights are constants), and the full training backward (data + weight gradients)."""
    import torch
    from torch.utils.flop_counter import FlopCounterMode
    from depthmodelhardening_amd import networks
    from depthmodelhardening_amd.depth_model import DepthModelWrapper
    with torch.device("meta"):      # built from the networks directly: import_depth_model() refuses any size but 1024x320
        enc = networks.ResnetEncoder(18, False)
        model = DepthModelWrapper(enc, networks.DepthDecoder(num_ch_enc=enc.num_ch_enc, scales=range(4)))
    model.eval()
    x = torch.empty(1, 3, height, width, device="meta", requires_grad=True)
    with FlopCounterMode(display=False) as fc:
        y = model(x)
    fwd = fc.get_total_flops()
    for p in model.parameters():
        p.requires_grad_(False)
    y = model(x)
    with FlopCounterMode(display=False) as fc:
        y.sum().backward()
    bwd_data = fc.get_total_flops()
    for p in model.parameters():
        p.requires_grad_(True)
    model.train()
    y = model(torch.empty(1, 3, height, width, device="meta"))
    with FlopCounterMode(display=False) as fc:
        y.sum().backward()
    return {"fwd": fwd, "bwd_data": bwd_data, "bwd_full": fc.get_total_flops()}


class FlopMeter(object):
    """Counts the U-Net passes of one step by hooking every ResnetEncoder: images x (forward [+ backward to the input if
    the weights are frozen (attack), + full backward if they train]) -> FLOPs per step with unet_flops()."""

    def __init__(self, modules):
        import torch
        from depthmodelhardening_amd import ops
        self.images = {"fwd_only": 0, "fwd_bwd_data": 0, "fwd_bwd_full": 0}
        self.on = False

        def hook(mod, args):
            if not self.on:
                return
            x = args[0]
            if not torch.is_grad_enabled():
                kind = "fwd_only"
            elif ops.weights_frozen() or not any(p.requires_grad for p in mod.parameters()):
                kind = "fwd_bwd_data" if x.requires_grad else "fwd_only"
            else:
                kind = "fwd_bwd_full"
            self.images[kind] += int(x.shape[0])
        self.handles = [m.register_forward_pre_hook(hook) for m in modules]

    def flops(self, per_image):
        n = self.images
        return (n["fwd_only"] * per_image["fwd"] + n["fwd_bwd_data"] * (per_image["fwd"] + per_image["bwd_data"]) +
                n["fwd_bwd_full"] * (per_image["fwd"] + per_image["bwd_full"]))


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
    """Start ``n`` rank processes of this script (one per GPU), watch ALL of them, relay rank 0's stdout.  If any rank
    exits non-zero the others are terminated at once (they would otherwise sit in the rendezvous or a collective until the
    process-group timeout) and the launcher exits non-zero: a fresh start is the only valid retry.  Returns the exit code."""
    import tempfile
    port = int(os.environ.get("MASTER_PORT") or _free_port())
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DMH_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = out0 if r == 0 else subprocess.DEVNULL          # stderr of every rank goes to the launcher's stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out))
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                rc = max(rc, abs(code))
                print("[bench launcher] rank %d exited with %d: stopping the other ranks" % (r, code), file=sys.stderr, flush=True)
                for q in live:
                    procs[q].terminate()
                deadline = time.time() + 10
                for q in list(live):
                    try:
                        procs[q].wait(timeout=max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        procs[q].kill()
                        procs[q].wait()
                live.clear()
                break
        if live:
            time.sleep(0.2)
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    return rc


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=sorted(BASELINE_CONFIGS),
                    help="BASELINE.json configs[N-1]: sets the flags of that workload (2 = the headline metric's)")
    ap.add_argument("--batch_size", type=int, default=32, help="per-GPU batch (weak scaling)")
    ap.add_argument("--global_batch", type=int, default=0,
                    help="strong scaling: total batch, split evenly over the ranks (SURVEY 8d: global 32 = 32/N per GPU)")
    ap.add_argument("--height", type=int, default=320)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--atk_steps", type=int, default=10)
    ap.add_argument("--atk_scenes", type=int, default=12, help="attack scenes per iteration (--atk_batch_size; reference: 12)")
    ap.add_argument("--shared_patch", action="store_true",
                    help="ONE patch for the job: the --atk_scenes scenes are sharded over the ranks, the patch gradient summed")
    ap.add_argument("--norm_type", type=str, default="l_inf", choices=["l_inf", "l_0"])
    ap.add_argument("--sync_attack", action="store_true")
    # the other BASELINE.json configs (parity/regression cases, not the headline line)
    ap.add_argument("--supervised_adv", action="store_true")
    ap.add_argument("--contrastive_learning", action="store_true")
    ap.add_argument("--loss_variant", type=str, default="md2", choices=["md2", "dh"])
    ap.add_argument("--harness", type=str, default="trainer", choices=["trainer", "physical"],
                    help="physical = BASELINE config 5: the physical_adv_training.py hardening loop (patch attack)")
    ap.add_argument("--graph_attack", action="store_true",
                    help="replay steps 2 .. n-1 of the L_inf attack from a HIP graph of step 1 (trainer --graph_attack)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--device", type=str, default="cuda", choices=["cuda", "cpu"],
                    help="cpu: launcher / collective plumbing test only (gloo); no kernels run")
    ap.add_argument("--phases", action="store_true", help="also print a per-phase GPU-time breakdown to stderr")
    a = ap.parse_args(argv)
    explicit = {x.split("=")[0].lstrip("-") for x in (sys.argv[1:] if argv is None else argv) if x.startswith("--")}
    for k, v in BASELINE_CONFIGS[a.config][1].items():
        if k not in explicit:           # an explicit flag wins over the config's preset
            setattr(a, k, v)
    return a


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
    if os.environ.get("DMH_BENCH_FAIL_RANK") == os.environ.get("RANK", "0"):     # test hook: this rank dies before the rendezvous
        raise SystemExit(3)
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
    import random
    random.seed(1234 + rank)        # the (z0, alpha) pose draws of the attack: the same run gives the same final_loss
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

    scaling = "weak"
    if a.global_batch:
        if a.global_batch % world:
            raise SystemExit("bench.py: --global_batch %d is not a multiple of %d ranks" % (a.global_batch, world))
        a.batch_size, scaling = a.global_batch // world, "strong"
    if a.harness == "physical":
        from depthmodelhardening_amd import physical_adv_training as pat
        job = pat.BenchJob(a.batch_size, a.atk_steps, rank, world, device)
        groups = [min(pat.MAX_POSE_GROUP, a.batch_size - lo) for lo in range(0, a.batch_size, pat.MAX_POSE_GROUP)]
        workload = ("physical_adv_training.py hardening loop: ONE Phy_obj_atk (EOT patch attack, %d steps) over all %d "
                    "scenes/GPU (poses drawn without replacement per group: %s) -> frozen model disparity -> MSE -> Adam, "
                    "Monodepth2 ResNet18 %dx%d" % (a.atk_steps, a.batch_size, "+".join(map(str, groups)), a.width, a.height))
    else:
        argv = ["--dataset", "synthetic", "--frame_ids", "0", "--use_stereo", "--height", str(a.height), "--width",
                str(a.width), "--batch_size", str(a.batch_size), "--learning_rate", "1e-5", "--adv_train", "--norm_type",
                a.norm_type, "--atk_steps", str(a.atk_steps), "--weights_init", "scratch", "--model_name", "bench",
                "--log_dir", os.path.join("/tmp", "dmh_bench_%d" % rank), "--synthetic_len", "1000000"]
        if a.sync_attack:
            argv.append("--sync_attack")
        if a.shared_patch:
            argv.append("--shared_patch")
        argv += ["--atk_batch_size", str(a.atk_scenes)]
        if a.graph_attack:
            argv.append("--graph_attack")
        if a.supervised_adv:
            argv.append("--supervised_adv")
        if a.contrastive_learning:
            argv.append("--contrastive_learning")
        argv += ["--loss_variant", a.loss_variant]
        opts = MonodepthOptions().parse(argv)
        job = Trainer(opts, rank=rank, world_size=world, device=device)
        job.set_train()
        workload = ("Monodepth2 ResNet18 %dx%d, %d-step %s attack on %d scenes%s, train batch %d/GPU, stereo "
                    "photometric+SSIM+smoothness loss (%s)%s%s, Adam" % (
                        a.width, a.height, a.atk_steps, "PGD-L_inf" if a.norm_type == "l_inf" else "L0/Adam", a.atk_scenes,
                        " (ONE patch: scenes sharded over the ranks)" if a.shared_patch and world > 1 else "",
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
    if world > 1:       # start-up (rendezvous, MIOpen's first-run compiles) is over: a collective stuck for minutes is a hang
        from depthmodelhardening_amd.ddp import shorten_timeout
        shorten_timeout()
        # test hooks (tests/test_gpu_bench_ranks.py): "exit:R" rank R dies here, "hang:R" rank R stops taking part -- the
        # others then sit in the next collective until the shortened timeout fails them
        hook = os.environ.get("DMH_BENCH_AFTER_WARMUP", "")
        if hook == "exit:%d" % rank:
            os._exit(3)
        if hook == "hang:%d" % rank:
            while True:
                time.sleep(1.0)
    # Inside the timed region only the K1 launches (the kernels of the roofline entry) carry HIP events: an event pair costs
    # ~10 us of dispatch latency around a launch, and a step has ~1,300 instrumented launches (timing all of them cost
    # 8.5 ms of a 174 ms step).  The other kernels' durations (roofline.others) come from ONE extra, untimed, fully
    # instrumented step BEFORE the timed region (so that the timed region is the tail of a kernel trace).
    from depthmodelhardening_amd.networks import ResnetEncoder
    holders = list(job.models.values()) + [getattr(job, "gt_model", None)] if a.harness == "trainer" else \
        [job.model_rob, job.model_ori]
    encoders = {id(m): m for h in holders if h is not None for m in h.modules() if isinstance(m, ResnetEncoder)}
    meter = FlopMeter(list(encoders.values()))
    ops.enable_profile(True)
    meter.on = True
    job.train_step()
    job._apply_pending_update()
    sync()
    meter.on = False
    kbytes = ops.profile_bytes()
    hot = ("photo_",) if a.harness == "trainer" else ("paste_",)      # the kernels of the roofline entry carry events
    ops.enable_profile(True, only=hot)
    atk = getattr(getattr(job, "dataset", None), "depth_atk", None) or getattr(job, "depth_atk", None)
    it0 = (getattr(atk, "total_iterations", 0), getattr(atk, "total_calls", 0))
    bucket = getattr(job, "bucket", None)
    if world > 1 and bucket is not None:
        bucket.timing = True
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses = job.train_step()
    job._apply_pending_update()
    sync()
    elapsed = time.perf_counter() - t0
    it1 = (getattr(atk, "total_iterations", 0), getattr(atk, "total_calls", 0))
    # the L0 attack runs atk_steps ... 2 atk_steps iterations by its patch's L0 ratio: the timed region's mean per attack
    atk_iters = round((it1[0] - it0[0]) / float(it1[1] - it0[1]), 2) if it1[1] > it0[1] else None
    note("timed region done: %.3fs for %d steps" % (elapsed, a.steps))
    kms = {k: v for k, v in ops.profile_ms().items() if k.startswith(hot)}
    kb_timed = ops.profile_bytes()
    ops.enable_profile(False)
    other_mode = None
    ar_times = None
    if world > 1 and bucket is not None:
        # the side-stream all-reduce of every timed step by HIP events (this rank's; the device is idle: sync() above)
        tm = bucket.timings()
        bucket.timing = False
        if tm["all_reduce_ms"]:
            n_ar = len(tm["all_reduce_ms"])
            ar_times = {"launches": n_ar, "all_reduce_ms_per_step": round(sum(tm["all_reduce_ms"]) / n_ar, 4),
                        "all_reduce_ms_max": round(max(tm["all_reduce_ms"]), 4),
                        "optimizer_wait_ms_per_step": round(sum(tm["wait_ms"]) / n_ar, 4),
                        "optimizer_wait_ms_max": round(max(tm["wait_ms"]), 4), "bucket_mb": round(bucket.numel * 4 / 1e6, 1),
                        "backend": dist.get_backend(),
                        "note": "HIP events on rank 0: all_reduce = the collective + 1/N scaling on the side stream, from its "
                                "first instruction to its last (stretched when it shares the CUs with the attack); optimizer_wait "
                                "= how long the compute stream stood at the collective's event before Adam (0 = fully hidden)"}
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if a.harness == "trainer":
            # the same K steps once more in the OTHER ordering of attack and gradient exchange (both numbers in one line):
            # overlapped = the attack is enqueued before the optimiser waits for the all-reduce (weights one step stale,
            # north_star); sync_attack = the reference's strict order (results identical to the reference's)
            job.opt.sync_attack = not job.opt.sync_attack
            job.train_step()
            job._apply_pending_update()
            sync()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                job.train_step()
            job._apply_pending_update()
            sync()
            t = torch.tensor([time.perf_counter() - t1], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            other_mode = {"mode": "sync_attack" if job.opt.sync_attack else "overlapped",
                          "value": round(a.batch_size * world * a.steps / float(t.item()), 3),
                          "ms_per_step": round(float(t.item()) / a.steps * 1e3, 3)}
            job.opt.sync_attack = not job.opt.sync_attack
    loss_val = float(losses["loss"].detach())

    if a.phases and rank == 0 and a.harness == "trainer":
        phase_breakdown(job)

    if rank == 0:
        fwd_b, bwd_b = k1_bytes(a.batch_size, a.height, a.width)
        names = {"photo_fwd": ("photo_fwd_kernel", fwd_b), "photo_bwd": ("photo_bwd_kernel", bwd_b)}
        if a.harness != "trainer":
            # physical_adv_training has no photometric loss: its hot-path kernel is K3 (EOT paste); launches differ in
            # size (32-scene attack steps, the two final pastes), so bytes and time are totals over the timed region
            names = {k: (k.replace("paste_", "paste_") + "_kernel", v[2] / max(1, v[0])) for k, v in kb_timed.items()
                     if k.startswith("paste_")}
            kms = {k: kb_timed[k][1] / max(1, kb_timed[k][0]) for k in names}
        dom = max(kms, key=lambda k: kms[k]) if kms else None
        roof = None
        if dom:
            kname, nbytes = names[dom]
            achieved = nbytes / (kms[dom] * 1e-3) / 1e9
            meas = measured_traffic(a.batch_size, a.height, a.width) if a.harness == "trainer" else {}
            traffic = meas.get(dom)
            roof = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                    "traffic_over_algorithmic": round(traffic / nbytes, 3) if traffic else None,
                    "traffic_source": meas.get("source"),
                    "note": ("algorithmic bytes = SURVEY 8d fused-variant figure; the fused loss does ~1,000 VALU instructions "
                             "per pixel on 30 B of traffic, so it sits on the VALU-issue roof, far below the HBM roof "
                             "(profiles/README.md)") if a.harness == "trainer" else
                            "K3 EOT paste (the hot-path kernel of this harness): SURVEY 8d bytes per launch, averaged over the "
                            "launches of the timed region",
                    "avg_ms": round(kms[dom], 4), "algorithmic_bytes": nbytes,
                    "traffic_stale": meas.get("stale"),
                    "valu_roof": valu_roof(meas[dom + "_valu_insts"], kms[dom], meas.get(dom + "_gui_active"))
                    if meas.get(dom + "_valu_insts") else None,
                    "others": {names[k][0]: {"avg_ms": round(v, 4), "GB/s": round(names[k][1] / (v * 1e-3) / 1e9, 1),
                                             "algorithmic_bytes": names[k][1], "traffic": meas.get(k),
                                             "valu_roof": valu_roof(meas[k + "_valu_insts"], v, meas.get(k + "_gui_active"))
                                             if meas.get(k + "_valu_insts") else None}
                               for k, v in kms.items() if k != dom}}
            # streaming kernels of the decoder glue: shapes vary per launch, so total bytes / total time
            for k, (cnt, ms, nb, fl) in sorted(kbytes.items()):
                if not k.startswith("photo_") and k not in names and ms > 0:
                    ent = {"launches_per_step": cnt, "ms_per_step": round(ms, 3)}      # the one instrumented step
                    if k.startswith("stem_conv_bwd"):      # K12 (both forms): 147 FMAs per gradient channel and 2x2 pixel block
                        tf = fl / (ms * 1e-3) / 1e12        # on the VECTOR ALU with the filter in SGPRs -- not on the MFMA
                        ent.update({"bound": "valu", "achieved": round(tf, 1), "peak": VALU_F32_SCALAR_TFLOPS, "unit": "TFLOP/s",
                                    "frac": round(tf / VALU_F32_SCALAR_TFLOPS, 4),
                                    "note": "peak = the MEASURED issue rate of scalar fp32 FMAs, 4.44 cycles per wave-instruction "
                                            "and SIMD (profiles/r04_valu_rate.txt); the data sheet's 157.3 TFLOP/s is the packed rate"})
                    elif fl > 0 and k.startswith("wino"):  # the Winograd-MFMA convolution (K10): bound by the fp32 matrix pipe
                        direct = fl / (ms * 1e-3) / 1e12
                        ent.update({"bound": "mfma", "TFLOP/s_direct_equivalent": round(direct, 1),
                                    "achieved": round(direct / 2.25, 1), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": round(direct / 2.25 / MFMA_F32_PEAK_TFLOPS, 4),
                                    "note": "achieved = MFMA flops actually issued (Winograd F(2x2,3x3): direct / 2.25)"})
                    elif fl > 0:                            # direct MFMA convolutions (K11 / K14 / K15 / K16): flops as counted
                        tf = fl / (ms * 1e-3) / 1e12
                        ent.update({"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4)})
                    else:
                        ent.update({"bound": "hbm", "GB/s": round(nb / (ms * 1e-3) / 1e9, 1),
                                    "frac": round(nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)})
                    roof["others"][k + "_kernel"] = ent
        # whole-step compute fraction: U-Net FLOPs per image measured with torch.utils.flop_counter, times the
        # U-Net passes counted in the instrumented step (direct-convolution FLOPs: the Winograd kernels issue 2.25x fewer)
        if roof is not None:
            try:
                per_image = unet_flops(a.height, a.width)
                ref_flops = meter.flops(per_image)
                # what the instrumented launches of one step actually computed (direct-convolution-equivalent FLOPs of the
                # hand-written convolution kernels: inside an attack the decoder tail and the encoder head's backward run on
                # windows around the object, K19); the few convolutions left to MIOpen are not counted
                done_flops = sum(fl for k, (cnt, ms, nb, fl) in kbytes.items())
                t_step = elapsed / a.steps
                roof["step"] = {"bound": "mfma", "flops_per_step": done_flops, "TFLOP/s": round(done_flops / t_step / 1e12, 2),
                                "peak": MFMA_F32_PEAK_TFLOPS, "frac": round(done_flops / t_step / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                                "reference_flops_per_step": ref_flops,
                                "reference_equivalent_TFLOP/s": round(ref_flops / t_step / 1e12, 2),
                                "unet_gflop_per_image": {k: round(v / 1e9, 3) for k, v in per_image.items()},
                                "unet_images_per_step": dict(meter.images),
                                "note": "flops_per_step = direct-convolution FLOPs the step's instrumented convolution launches "
                                        "executed (Winograd kernels issue 1/2.25 of them on the MFMA), / measured step time, against "
                                        "the dense fp32 MFMA peak; reference_flops_per_step = the same step with every U-Net pass "
                                        "over the whole frame, as the reference runs it (torch.utils.flop_counter on the module "
                                        "path) -- the attack's windows (K19) skip the difference, it is not work done"}
            except Exception as e:      # a reporting extra must never cost the JSON line (or the barrier behind it)
                roof["step"] = None
                note("roofline.step unavailable: %r" % (e,))
        steps_txt = "%d-step %s" % (a.atk_steps, "PGD" if a.norm_type == "l_inf" else "L0")
        metric = "adv-train images/sec @%dx%d, %s, bs%d" % (a.width, a.height, steps_txt, a.batch_size)
        if a.atk_scenes != 12 or a.shared_patch:
            metric += ", %d attack scenes%s" % (a.atk_scenes, " (shared patch)" if a.shared_patch else "")
        if a.harness != "trainer":
            metric = "physical_adv_training images/sec @%dx%d, %s patch attack, bs%d" % (a.width, a.height, steps_txt, a.batch_size)
        out = {"metric": metric, "value": round(a.batch_size * world * a.steps / elapsed, 3),
               "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": scaling,
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": workload, "baseline_config": BASELINE_CONFIGS[a.config][0], "config_index": a.config,
                          "global_batch": a.batch_size * world, "per_gpu_batch": a.batch_size, "parallelism": "dp%d" % world,
                          "attack_overlap": bool(world > 1 and not a.sync_attack and a.harness == "trainer"),
                          "attack_scenes": a.atk_scenes, "shared_patch": bool(a.shared_patch),
                          "graph_attack": bool(a.graph_attack), "final_loss": round(loss_val, 6)},
               "roofline": roof}
        if atk_iters is not None:
            out["config"]["attack_iterations_per_step"] = atk_iters
        if other_mode is not None:
            out["config"]["other_order"] = other_mode
        if ar_times is not None:
            out["config"]["all_reduce"] = ar_times
        if world == 1 and not a.no_cpu_baseline:
            gpu_loss_ms = round(sum(kms.values()), 4) if kms else None
            out["cpu_baseline"] = cpu_baseline(a.height, a.width, a.atk_steps, a.batch_size,
                                               gpu_loss_ms if a.harness == "trainer" else None)
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
