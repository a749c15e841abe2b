"""depthmodelhardening_amd -- MI355X-native adversarial-training hot path for self-supervised depth
hardening (Monodepth2 / DepthHints): hand-written gfx950 HIP kernels behind a C ABI
(include/dmh_hip.h, csrc/), and the host-side mirror of the reference's Python surface
(Trainer.compute_losses, torchattacks.Phy_obj_atk*, PhysicalTrans, MonoDataset adv hooks)."""
__version__ = "0.1.0"
