"""Command-line options: every flag of the reference's ``MonodepthOptions`` (MD2/options.py:15-233) keeps
its spelling, type and default.  Added (marked NEW): the attack hyper-parameters that the reference
hard-codes in MD2/trainer.py:199-223, a synthetic dataset, and the data-parallel switches."""
import argparse
import os

file_dir = os.path.dirname(__file__)


class MonodepthOptions:
    def __init__(self):
        p = argparse.ArgumentParser(description="Monodepthv2 options (MI355X adversarial-training build)")
        # PATHS
        p.add_argument("--data_path", type=str, default=os.path.join(file_dir, "kitti_data"))
        p.add_argument("--log_dir", type=str, default=os.path.join(os.path.expanduser("~"), "tmp"))
        # TRAINING
        p.add_argument("--model_name", type=str, default="mdp")
        p.add_argument("--split", type=str, choices=["eigen_zhou", "eigen_full", "odom", "benchmark"],
                       default="eigen_zhou")
        p.add_argument("--num_layers", type=int, default=18, choices=[18, 34, 50, 101, 152])
        p.add_argument("--dataset", type=str, default="kitti",
                       choices=["kitti", "kitti_odom", "kitti_depth", "kitti_test", "synthetic"])
        p.add_argument("--png", action="store_true")
        p.add_argument("--height", type=int, default=192)
        p.add_argument("--width", type=int, default=640)
        p.add_argument("--disparity_smoothness", type=float, default=1e-3)
        p.add_argument("--scales", nargs="+", type=int, default=[0, 1, 2, 3])
        p.add_argument("--min_depth", type=float, default=0.1)
        p.add_argument("--max_depth", type=float, default=100.0)
        p.add_argument("--use_stereo", action="store_true")
        p.add_argument("--frame_ids", nargs="+", type=int, default=[0, -1, 1])
        p.add_argument("--adv_train", action="store_true", help="do adversarial training")
        p.add_argument("--fine_tune", action="store_true", help="do finetune on an existing model")
        p.add_argument("--supervised_adv", action="store_true", help="add the supervised loss on the adversarial view")
        # no default, as in the reference (MD2/options.py:94-96); Trainer.__init__ asks for it under --adv_train, where the
        # reference would run into a NameError at MD2/trainer.py:224 (neither branch assigns ``args``)
        p.add_argument("--norm_type", type=str, choices=["l_inf", "l_0"])
        # OPTIMIZATION
        p.add_argument("--batch_size", type=int, default=12)
        p.add_argument("--learning_rate", type=float, default=1e-4)
        p.add_argument("--num_epochs", type=int, default=20)
        p.add_argument("--scheduler_step_size", type=int, default=15)
        # ABLATION
        p.add_argument("--v1_multiscale", action="store_true")
        p.add_argument("--avg_reprojection", action="store_true")
        p.add_argument("--disable_automasking", action="store_true")
        p.add_argument("--predictive_mask", action="store_true")
        p.add_argument("--no_ssim", action="store_true")
        p.add_argument("--weights_init", type=str, default="pretrained", choices=["pretrained", "scratch"])
        p.add_argument("--pose_model_input", type=str, default="pairs", choices=["pairs", "all"])
        p.add_argument("--pose_model_type", type=str, default="separate_resnet",
                       choices=["posecnn", "separate_resnet", "shared"])
        p.add_argument("--contrastive_learning", action="store_true")
        p.add_argument("--no_original_train", action="store_true")
        p.add_argument("--half_no_synthesis", action="store_true")
        # SYSTEM
        p.add_argument("--no_cuda", action="store_true")
        p.add_argument("--num_workers", type=int, default=12)
        # LOADING
        p.add_argument("--load_weights_folder", type=str)
        p.add_argument("--models_to_load", nargs="+", type=str, default=["encoder", "depth", "pose_encoder", "pose"])
        # LOGGING
        p.add_argument("--log_frequency", type=int, default=250)
        p.add_argument("--save_frequency", type=int, default=1)
        # EVALUATION (accepted for command-line compatibility; evaluation scripts are out of scope)
        p.add_argument("--eval_stereo", action="store_true")
        p.add_argument("--eval_mono", action="store_true")
        p.add_argument("--disable_median_scaling", action="store_true")
        p.add_argument("--pred_depth_scale_factor", type=float, default=1)
        p.add_argument("--ext_disp_to_eval", type=str)
        p.add_argument("--eval_split", type=str, default="eigen",
                       choices=["eigen", "eigen_benchmark", "benchmark", "odom_9", "odom_10"])
        p.add_argument("--save_pred_disps", action="store_true")
        p.add_argument("--no_eval", action="store_true")
        p.add_argument("--eval_eigen_to_benchmark", action="store_true")
        p.add_argument("--eval_out_dir", type=str)
        p.add_argument("--post_process", action="store_true")
        p.add_argument("--gt_depth", action="store_true")
        # NEW: attack hyper-parameters (defaults = the values hard-coded at MD2/trainer.py:199-223)
        p.add_argument("--atk_steps", type=int, default=10)
        p.add_argument("--atk_eps", type=float, default=0.1)
        p.add_argument("--atk_alpha", type=float, default=0.02)
        p.add_argument("--atk_batch_size", type=int, default=12)
        p.add_argument("--atk_adam_lr", type=float, default=0.5)
        p.add_argument("--atk_mask_wt", type=float, default=0.06)
        p.add_argument("--atk_l0_thresh", type=float, default=0.1)
        # NEW: variant / data / distribution
        p.add_argument("--loss_variant", type=str, default="md2", choices=["md2", "dh"],
                       help="md2 = MD2/trainer.py:647-660; dh = DepthHints normalisation, DH/trainer.py:700-708")
        p.add_argument("--synthetic_len", type=int, default=64, help="items per epoch of --dataset synthetic")
        p.add_argument("--seed", type=int, default=1234)
        p.add_argument("--sync_attack", action="store_true",
                       help="strict reference order: wait for the gradient all-reduce + Adam before the next attack")
        p.add_argument("--shared_patch", action="store_true",
                       help="data-parallel runs: ONE patch for the whole job, as the reference's single process has (MD2/"
                            "trainer.py:300-307) -- the --atk_batch_size attack scenes are sharded over the ranks and the patch "
                            "gradient is summed over them before every sign step (SURVEY.md section 8e).  Default: every rank "
                            "attacks its own --atk_batch_size scenes and keeps its own patch (no attack-time communication)")
        p.add_argument("--materialize_warps", action="store_true",
                       help="generate_images_pred also writes depth/sample/color tensors (the fused loss never reads them)")
        p.add_argument("--use_depth_hints", action="store_true",
                       help="DepthHints (DH/options.py:99-101): depth-hint reprojection candidate + proxy supervision; needs "
                            "--loss_variant dh.  The synthetic dataset makes up hints (smooth depth with holes)")
        p.add_argument("--reference_stale_patch", action="store_true",
                       help="paste the adversarial patch as of the epoch start, as the reference's forked DataLoader workers "
                            "do (SURVEY.md section 3.1); default: the freshly attacked patch (its num_workers=0 behaviour)")
        p.add_argument("--no_flip_sides", action="store_true",
                       help="synthetic data: always camera side 'l' and no horizontal flips")
        p.add_argument("--max_steps", type=int, default=0, help="stop after this many iterations (0 = full epochs)")
        p.add_argument("--graph_attack", action="store_true",
                       help="L_inf attack: one set of window sizes for all steps, step 1 captured in a HIP graph and replayed "
                            "for steps 2 .. n-1 (torchattacks/attacks/phy_obj_atk.py, use_graph): takes the step's ~130 launches "
                            "off the host; for ranks whose GPU share is small (strong scaling)")
        p.add_argument("--step_log", type=str, default="",
                       help="JSONL step log written by rank 0 (SURVEY.md section 5): one line per iteration with the loss, "
                            "images/s and the GPU time of each phase (attack / forward + loss / backward / all-reduce + "
                            "Adam) from HIP events, read one iteration late so that logging never drains the queue")
        self.parser = p

    def parse(self, args=None):
        self.options = self.parser.parse_args(args)
        return self.options
