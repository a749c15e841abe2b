"""Experiment: how much of the step depends on the SIZE of the attack windows?  Runs bench.py's step with the object's distance
range narrowed (all objects far = small windows, all near = large windows) beside the reference's range (5 ... 9.8 m, where the
twelve windows of a step share the largest object's size).  Bounds what per-scene window sizes could gain.
    python tools/exp_window_sizes.py far|near|ref [bench.py flags]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
which = sys.argv.pop(1)
import depthmodelhardening_amd.datasets.synthetic as synth      # noqa: E402

rng = {"far": np.arange(9.0, 9.96, 0.08), "near": np.arange(5.0, 5.96, 0.08), "ref": np.arange(5, 10, 0.2)}[which]
synth.train_dist_range = list(rng)
import bench                                                      # noqa: E402

if __name__ == "__main__":
    bench.main()
