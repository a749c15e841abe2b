"""Two ranks on ONE MI355X over gloo (RCCL needs one GPU per rank; the driver runs the real multi-GPU bench):
exercises the Trainer's data-parallel schedule end to end -- parameter broadcast, flat gradient bucket, side-stream
all-reduce, attack/optimiser ordering in both the overlapped and the --sync_attack mode."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


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
            "--model_name", "t", "--synthetic_len", "8", "--adv_train", "--atk_steps", "1", "--atk_batch_size", "2"]
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
