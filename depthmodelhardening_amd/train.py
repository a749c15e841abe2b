"""Entry point: ``python -m depthmodelhardening_amd.train --adv_train --norm_type l_inf ...`` -- the
reference's ``MD2/train.py:12-18`` (options -> Trainer -> train()), one process per GPU under torchrun."""
from .ddp import init_distributed
from .options import MonodepthOptions
from .trainer import Trainer


def main(argv=None):
    opts = MonodepthOptions().parse(argv)
    rank, world, device = init_distributed("cpu" if opts.no_cuda else "cuda")
    trainer = Trainer(opts, rank=rank, world_size=world, device=device)
    trainer.train()


if __name__ == "__main__":
    main()
