"""Two ranks on ONE MI355X over gloo (RCCL needs one GPU per rank; the driver runs the real multi-GPU bench):
exercises the Trainer's data-parallel schedule end to end -- parameter broadcast, flat gradient bucket, side-stream
all-reduce, attack/optimiser ordering in both the overlapped and the --sync_attack mode."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

# one xdist group: these tests start rank processes of their own, and the box allows six processes on the GPU
pytestmark = [pytest.mark.gpu, pytest.mark.xdist_group("ranks")]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp, sync_attack, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", DMH_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from depthmodelhardening_amd.ddp import init_distributed
    from depthmodelhardening_amd.options import MonodepthOptions
    from depthmodelhardening_amd.trainer import Trainer
    r, w, dev = init_distributed("cuda")
    torch.manual_seed(100 + rank)                      # different initial weights per rank: broadcast must fix it
    argv = ["--dataset", "synthetic", "--frame_ids", "0", "--use_stereo", "--height", "64", "--width", "192",
            "--batch_size", "2", "--weights_init", "scratch", "--log_dir", os.path.join(tmp, "r%d" % rank),
            "--model_name", "t", "--synthetic_len", "8", "--adv_train", "--norm_type", "l_inf", "--atk_steps", "1", "--atk_batch_size",
            "2"]
    if sync_attack:
        argv.append("--sync_attack")
    tr = Trainer(MonodepthOptions().parse(argv), rank=r, world_size=w, device=dev)
    tr.set_train()
    w0 = torch.cat([p.detach().reshape(-1) for p in tr.models["depth"].parameters()]).cpu()
    for _ in range(3):
        losses = tr.train_step()
    tr._apply_pending_update()
    torch.cuda.synchronize()
    w1 = torch.cat([p.detach().reshape(-1) for p in tr.models["depth"].parameters()]).cpu()
    ret[rank] = (w0, w1, float(losses["loss"].detach()), tr.bucket.numel)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sync_attack", [False, True])
def test_two_ranks_stay_in_lockstep(tmp_path, sync_attack):
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), sync_attack, ret), nprocs=2, join=True)
    (a0, a1, la, na), (b0, b1, lb, nb) = ret[0], ret[1]
    assert torch.equal(a0, b0), "broadcast_parameters did not equalise the initial weights"
    assert torch.equal(a1, b1), "ranks diverged after three data-parallel steps"
    assert not torch.equal(a0, a1) and torch.isfinite(a1).all()
    assert la != lb                                   # each rank trains on its own shard
    assert na == nb == 14329236                       # encoder (without fc) + decoder: the 57.3 MB bucket


def _nccl_worker(rank, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      DMH_DIST_FORCE_INIT="1")
    os.environ.pop("DMH_DIST_BACKEND", None)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from depthmodelhardening_amd import networks
    from depthmodelhardening_amd.ddp import GradBucket, init_distributed
    r, w, dev = init_distributed("cuda")
    assert dist.is_initialized() and dist.get_backend() == "nccl" and w == 1
    torch.manual_seed(3)
    enc = networks.ResnetEncoder(18, False).to(dev)
    dec = networks.DepthDecoder(enc.num_ch_enc, range(4)).to(dev)
    fc = {id(p) for p in enc.encoder.fc.parameters()}
    bucket = GradBucket([p for m in (enc, dec) for p in m.parameters() if id(p) not in fc], world_size=1, force_collective=True)
    assert bucket.numel == 14329236 and bucket.stream is not None
    # gradients are produced on the compute stream by a long chain of kernels; the collective on the side stream must wait
    # for them (wait_stream), and the optimiser's stream for the collective (event)
    g = torch.Generator(device=dev).manual_seed(5)
    want = torch.randn(bucket.numel, device=dev, generator=g)
    bucket.zero()
    x = torch.zeros_like(bucket.flat)
    for _ in range(50):                         # ~50 x 57 MB of element-wise work queued ahead of the producer
        x.add_(0.0)
    bucket.flat.copy_(want + x)                 # the "backward": written last on the compute stream (x == 0: exact)
    bucket.start_all_reduce()                   # RCCL all-reduce of the 57.3 MB bucket on the side stream
    assert bucket._event is not None
    bucket.finish_all_reduce()
    got = bucket.flat.clone()                   # ordered after the event on the compute stream
    torch.cuda.synchronize()
    same = bool(torch.equal(got, want))         # sum over one rank / 1: bitwise unchanged
    bucket.check_attached()
    views_ok = all(p.grad.data_ptr() == bucket.flat.data_ptr() + off * 4 for p, off in zip(bucket.params, bucket._offsets))
    # a second round trip, as the training loop does every iteration
    bucket.flat.mul_(2.0)
    bucket.all_reduce()
    torch.cuda.synchronize()
    same2 = bool(torch.equal(bucket.flat, want * 2.0))
    ret[0] = (same, views_ok, same2)
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_path_runs_once_with_one_rank():
    """Backend ``nccl`` (= RCCL) initialised with world_size 1: the 57.3 MB bucket goes through start / finish_all_reduce
    on the side stream exactly as on an 8-GPU node -- RCCL is loaded, a communicator is built, the collective kernel runs,
    stream / event ordering holds and the gradients come back bitwise unchanged (sum over one rank, divided by one)."""
    ret = mp.Manager().dict()
    mp.spawn(_nccl_worker, args=(_free_port(), ret), nprocs=1, join=True)
    same, views_ok, same2 = ret[0]
    assert same and same2, "the gradients changed on their way through the 1-rank RCCL all-reduce"
    assert views_ok
