"""Host-side logic of the product (no GPU): geometry, coefficient solve, RNG draw order, options, isolation."""
import os
import random
import re

import numpy as np
import pytest
import torch

from depthmodelhardening_amd import my_utils
from depthmodelhardening_amd.options import MonodepthOptions
from depthmodelhardening_amd.physicalTrans import PhysicalTrans, get_perspective_coeffs, read_calib_P2
from oracle import attack_ref, synth, tv082

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_quads_match_reference_golden(golden):
    g = golden("geometry")
    obj, mask = synth.make_object()
    pt = PhysicalTrans(obj, mask, {"path": None}, (1, 3, 375, 1242), dist_range=my_utils.train_dist_range)
    assert pt.calib_source == "builtin:003086"
    assert np.array_equal(np.array(pt.pos_obj_img_start, dtype=np.int32), g["start"])
    adv_K = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
    adv_K[0, :] *= 1242
    adv_K[1, :] *= 375
    for i, z0 in enumerate(pt.dist_range):
        for j, al in enumerate(pt.angle_range):
            assert np.array_equal(pt.objPosOnImage(z0, al), g["quads"][i, j])
            assert np.array_equal(pt.objPosOnImage(z0, al, adv_K), g["quads_K"][i, j])


def test_calib_file_and_coeffs(tmp_path):
    p = tmp_path / "003086.txt"
    p.write_text(synth.KITTI_CALIB_TEXT)
    assert np.allclose(read_calib_P2(str(p)), my_utils.KITTI_003086_P2)
    obj, mask = synth.make_object()
    pt = PhysicalTrans(obj, mask, {"path": str(p)}, (1, 3, 375, 1242))
    assert pt.calib_source == str(p)
    ref = attack_ref.PhysicalTransRef(obj, mask)
    T = np.eye(4, dtype=np.float32)
    T[0, 3] = -0.54
    c = pt.coeffs_for([5.0, 9.4], [-30, 15])
    for i, (z0, al) in enumerate([(5.0, -30), (9.4, 15)]):
        want = tv082.get_perspective_coeffs([list(map(float, q)) for q in ref.pos_obj_img_start],
                                            [list(map(float, q)) for q in ref.obj_pos_on_image(z0, al)])
        assert np.allclose(c[i], np.array(want, dtype=np.float32), rtol=0, atol=0)
    assert np.array_equal(pt._objPosOnImage_w_trans(T, 6.0, 10), ref.obj_pos_on_image(6.0, 10, None, T))
    # identity quad -> identity homography
    start = [[float(v) for v in q] for q in pt.pos_obj_img_start]
    assert np.allclose(get_perspective_coeffs(start, start), [1, 0, 0, 0, 1, 0, 0, 0], atol=1e-6)
    try:
        PhysicalTrans(obj, mask, {"path": None}, (1, 3, 320, 1024))
        assert False
    except AssertionError:
        pass


def test_sample_draw_order_matches_reference_semantics():
    obj, mask = synth.make_object()
    pt = PhysicalTrans(obj, mask, {"path": None}, (1, 3, 375, 1242), dist_range=my_utils.train_dist_range)
    random.seed(3)
    z0, al = pt.draw_samples(12)
    random.seed(3)
    assert z0 == random.sample(pt.dist_range, 12) and al == random.sample(pt.angle_range, 12)
    assert len(set(al)) == 12      # without replacement: at most 13 samples (SURVEY a-10)
    try:
        pt.draw_samples(14)
        assert False, "14 angles cannot be drawn without replacement"
    except ValueError:
        pass


def test_options_keep_reference_flags_and_defaults():
    o = MonodepthOptions().parse([])
    assert (o.height, o.width, o.batch_size, o.learning_rate, o.num_epochs) == (192, 640, 12, 1e-4, 20)
    assert o.frame_ids == [0, -1, 1] and o.scales == [0, 1, 2, 3] and o.disparity_smoothness == 1e-3
    assert (o.min_depth, o.max_depth, o.num_workers, o.log_frequency) == (0.1, 100.0, 12, 250)
    assert (o.atk_steps, o.atk_eps, o.atk_alpha, o.atk_batch_size) == (10, 0.1, 0.02, 12)   # MD2/trainer.py:199-211
    assert (o.atk_adam_lr, o.atk_mask_wt, o.atk_l0_thresh) == (0.5, 0.06, 0.1)               # MD2/trainer.py:212-223
    p = MonodepthOptions().parse("--frame_ids 0 --use_stereo --split eigen_full --png --width 1024 --height 320 "
                                 "--learning_rate 1e-5 --adv_train --norm_type l_0 --contrastive_learning "
                                 "--supervised_adv --batch_size 32 --num_workers 8".split())
    assert p.adv_train and p.norm_type == "l_0" and p.contrastive_learning and p.supervised_adv and p.width == 1024


def test_constants():
    assert (my_utils.ori_H, my_utils.ori_W) == (375, 1242)
    assert len(my_utils.train_dist_range) == 25 and abs(my_utils.train_dist_range[-1] - 9.8) < 1e-9


def test_cpu_tensors_are_rejected_not_silently_computed():
    from depthmodelhardening_amd import ops
    x = torch.rand(4)
    try:
        ops.pgd_linf_step(x, x, x, 0.1, 0.1)
        assert False
    except RuntimeError as e:
        assert "no CPU path" in str(e)


def test_product_never_imports_the_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for root, _, files in os.walk(os.path.join(REPO, "depthmodelhardening_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                assert not pat.search(open(os.path.join(root, f)).read()), f


def test_trainer_on_cpu_fails_loudly_instead_of_falling_back(tmp_path):
    """Config 1 of BASELINE.json is the reference's CPU plumbing case: the networks run on the CPU, but the hot
    path has no CPU implementation in the product and must say so (the CPU side of parity is the oracle's job)."""
    from depthmodelhardening_amd.trainer import Trainer
    opts = MonodepthOptions().parse(["--dataset", "synthetic", "--frame_ids", "0", "--use_stereo", "--height", "64",
                                     "--width", "192", "--batch_size", "2", "--weights_init", "scratch", "--no_cuda",
                                     "--log_dir", str(tmp_path), "--model_name", "cpu", "--synthetic_len", "4"])
    tr = Trainer(opts, device=torch.device("cpu"), host_only=True)
    assert tr.bucket.numel == 14329236
    inputs = tr.dataset.next_batch(2)
    feats = tr.models["encoder"](inputs["color_aug", 0, 0])
    outputs = tr.models["depth"](feats)                      # decoder takes its reference (ATen) path on the CPU
    assert outputs[("disp", 0)].shape == (2, 1, 64, 192)
    try:
        tr.compute_losses(inputs, outputs)
        assert False, "compute_losses must not silently run an eager/CPU path"
    except RuntimeError as e:
        assert "no CPU path" in str(e)


def test_no_cuda_and_missing_norm_type_fail_in_the_constructor(tmp_path):
    """--no_cuda (MD2/options.py) is accepted on the command line and refused by Trainer.__init__ -- not inside the first attack
    step; --norm_type has no default (MD2/options.py:94-96) and --adv_train without it is an error with a message, where the
    reference runs into a NameError (MD2/trainer.py:224)."""
    from depthmodelhardening_amd.trainer import Trainer
    base = ["--dataset", "synthetic", "--frame_ids", "0", "--use_stereo", "--height", "64", "--width", "192", "--batch_size",
            "2", "--weights_init", "scratch", "--log_dir", str(tmp_path), "--model_name", "x", "--synthetic_len", "4"]
    assert MonodepthOptions().parse(base).norm_type is None
    with pytest.raises(RuntimeError, match="no CPU path"):
        Trainer(MonodepthOptions().parse(base + ["--no_cuda"]))
    with pytest.raises(RuntimeError, match="no CPU path"):
        Trainer(MonodepthOptions().parse(base), device=torch.device("cpu"))
    with pytest.raises(RuntimeError, match="--norm_type"):
        Trainer(MonodepthOptions().parse(base + ["--adv_train"]), device=torch.device("cpu"), host_only=True)


def test_step_log_reports_device_time_not_enqueue_time(tmp_path):
    """StepLog.images_per_s comes from the phase sum (HIP events; perf_counter marks on the CPU), not from the host's enqueue
    interval: a host that runs ahead of the device would otherwise report 2x the real rate (profiles/r05_steps.jsonl)."""
    import json
    import time
    from depthmodelhardening_amd.trainer import StepLog
    log = StepLog(str(tmp_path / "s.jsonl"), images_per_step=32)
    log.cuda = False                 # host clocks: this test has no GPU
    for step in range(3):
        log.mark("start")
        time.sleep(0.02)
        log.mark("attack")
        time.sleep(0.03)
        log.mark("backward")
        log.end_step(step, 0, 1.5)
    log.close()
    lines = [json.loads(l) for l in open(tmp_path / "s.jsonl")]
    assert len(lines) == 3
    for line in lines:
        total = sum(line["phase_ms"].values())
        assert 45.0 <= total <= 80.0, line
        assert abs(line["images_per_s"] - 32 / (total * 1e-3)) <= 0.02 * line["images_per_s"], line
        assert line["device_ms"] == pytest.approx(total, abs=0.01)


def test_the_library_reports_its_own_launch_plan():
    """dmh_wino_conv3x3_plan: what a K10 call of a shape would do (tile regions, work items, two-way split, stream-K), asked by
    ops._wino_ok instead of mirrored (pure host logic: no launch, no GPU)."""
    from depthmodelhardening_amd import _native as N, ops
    lib = N.lib()
    ws = 2 * 256 * 16384
    p = lib.dmh_wino_conv3x3_plan(12, 64, 64, 80, 256, 1, 1, ws)            # layer1, attack batch, fused epilogue
    assert p >= 0 and (p & 3) == 0 and ((p >> 2) & 3) == 0 and (p >> 8) == 12 * 20 * 4          # 960 whole items, 2 x 32 regions
    p = lib.dmh_wino_conv3x3_plan(12, 512, 256, 12, 34, 0, 0, ws)           # upconv(4,0): 60 regions of 4 x 16 tiles over the batch
    assert p >= 0 and (p & 1) == 1 and ((p >> 2) & 3) == 2 and (p >> 8) == 60
    p = lib.dmh_wino_conv3x3_plan(12, 512, 256, 12, 34, 0, 0, 0)            # ... without a workspace: the two-way channel split
    assert p >= 0 and (p & 3) == 2 and (p >> 8) == 120
    p = lib.dmh_wino_conv3x3_plan(12, 512, 512, 10, 32, 1, 1, ws)           # layer4 under the epilogue: few regions -> stream-K
    assert p >= 0 and (p & 1) == 1
    assert lib.dmh_wino_conv3x3_plan(12, 20, 64, 80, 256, 1, 0, ws) == -1   # input channels not a multiple of 8
    assert lib.dmh_wino_conv3x3_plan(12, 64, 64, 81, 256, 1, 0, ws) == -1   # odd output height
    assert ops._wino_ok(12, 512, 256, 10, 32) and ops._wino_ok(12, 512, 512, 10, 32, allow_split=False, allow_sk=True)


def test_conv_dispatch_rule_mirrors_the_launcher():
    """ops._wino_ok / _small_ok: which 3x3 convolutions of a step go to K10 / K11 (pure host logic)."""
    from depthmodelhardening_amd import ops
    B = 12
    assert ops._wino_ok(B, 64, 64, 80, 256)             # encoder layer1
    assert ops._wino_ok(B, 512, 512, 10, 32)            # layer4: 4x16 regions over the flattened batch + channel split
    assert not ops._wino_ok(B, 512, 512, 10, 32, allow_split=False) or ops._wino_ok(B, 512, 512, 10, 32)
    assert ops.WINO_SK and ops._wino_ok(B, 512, 256, 10, 32)   # upconv4_0 forward, 60 regions: stream-K deals its 3,840 (item,
    ops.WINO_SK = False                                         # chunk) units to 256 workgroups (round 5) ...
    try:
        assert not ops._wino_ok(B, 512, 256, 10, 32)            # ... whole items: too few of them (MIOpen, as until round 4)
    finally:
        ops.WINO_SK = True
    assert ops._wino_ok(B, 256, 512, 12, 34)            # its backward: 17 tile columns in a 32-wide region fill 53 % of the
    assert not ops._wino_ok(B, 256, 512, 10, 34)        # tiles (taken since round 5, 120 against MIOpen's 138 us); 44 %: MIOpen
    assert not ops._wino_ok(B, 96, 32, 160, 512)        # 32 output channels half-fill an item
    assert not ops._wino_ok(B, 32, 96, 162, 514)        # few chunks and a half-empty channel group
    assert not ops._wino_ok(B, 64, 64, 81, 256)         # odd output height
    assert not ops._wino_ok(B, 16, 64, 80, 256)         # < 24 input channels
    assert ops._small_ok(16, 16) and ops._small_ok(32, 16) and ops._small_ok(16, 32) and ops._small_ok(1, 16)
    assert not ops._small_ok(32, 32) and not ops._small_ok(64, 16)


def test_frozen_weights_scope_and_coefficient_memo():
    import numpy as np
    import torch
    from depthmodelhardening_amd import ops
    from depthmodelhardening_amd.my_utils import object_dataset_root, ori_H, ori_W, to_device_async, train_dist_range
    from depthmodelhardening_amd.physicalTrans import PhysicalTrans
    calls = []
    assert ops.frozen_memo("k", lambda: calls.append(1) or 5) == 5 and ops.frozen_memo("k", lambda: calls.append(1) or 5) == 5
    assert len(calls) == 2                               # outside a scope nothing is cached
    with ops.frozen_weights():
        with ops.frozen_weights():                       # nests
            assert ops.frozen_memo("k", lambda: calls.append(1) or 7) == 7
        assert ops.frozen_memo("k", lambda: calls.append(1) or 8) == 7
    assert len(calls) == 3 and not ops._wino_cache       # dropped on exit of the outermost scope
    t = to_device_async(np.arange(6, dtype=np.float32).reshape(2, 3), "cpu")
    assert t.shape == (2, 3) and to_device_async([1, 2], "cpu", torch.int64).dtype == torch.int64
    obj, mask = torch.rand(1, 3, 260, 300), torch.ones(1, 1, 260, 300)
    conf = {"path": f"{object_dataset_root}/training/calib/003086.txt"}
    pt = PhysicalTrans(obj, mask, conf, (1, 3, ori_H, ori_W), dist_range=train_dist_range)
    fresh = PhysicalTrans(obj, mask, conf, (1, 3, ori_H, ori_W), dist_range=train_dist_range)
    z, a = pt.draw_samples(6)
    c1 = pt.coeffs_for(z, a)                             # warms the whole pose grid, then memo hits
    c2 = pt.coeffs_for(z, a)
    np.testing.assert_array_equal(c1, c2)
    for i in range(len(z)):                              # single-sample calls on a fresh object never warm the grid
        np.testing.assert_array_equal(c1[i], fresh.coeffs_for([z[i]], [a[i]])[0])
    assert len(fresh._coeff_memo) <= len(z)
    assert len(pt._coeff_memo) >= len(pt.dist_range) * len(pt.angle_range)


def test_bench_presets_name_the_baseline_configs():
    """bench.py --config N must run the workload BASELINE.json configs[N-1] names, and the default
    metric string must be BASELINE.json's headline metric."""
    import json
    import os
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    import bench
    base = json.load(open(os.path.join(repo, "BASELINE.json")))
    norm = lambda t: t.replace("×", "x").replace("L∞", "L_inf").replace(" ", "")     # noqa: E731
    for idx, (name, flags) in bench.BASELINE_CONFIGS.items():
        assert norm(name) == norm(base["configs"][idx - 1]), (idx, name)
    a = bench.parse([])
    assert (a.config, a.batch_size, a.atk_steps, a.height, a.width, a.norm_type) == (2, 32, 10, 320, 1024, "l_inf")
    assert norm(base["metric"]).startswith(norm("adv-train images/sec @1024x320, 10-step PGD, bs32"))
    a4 = bench.parse(["--config", "4"])
    assert (a4.batch_size, a4.atk_steps, a4.loss_variant, a4.contrastive_learning) == (64, 20, "dh", True)
    # the step-FLOP model builds at any size (import_depth_model refuses everything but 1024x320)
    f = bench.unet_flops(64, 192)
    assert f["fwd"] > 0 and f["bwd_full"] > f["bwd_data"] > 0


def test_color_jitter_matches_the_torchvision_restatement():
    """color_jitter.get_params / its four operations (the product's closed-form HSV rotation) against oracle/tv082.py's op-for-op
    restatement of torchvision 0.8.2 (six-case table): the same draws from ``random`` in the same order, the same images, the
    same gradients -- over 20 random transforms, flat (grey) pixels and channel ties included."""
    import random
    import torch
    from depthmodelhardening_amd import color_jitter as cj
    from oracle import tv082
    x = torch.rand(2, 3, 24, 40, dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    x[0, :, :4] = 0.3                   # flat pixels: max == min
    x[1, 0, 4:8] = x[1, 1, 4:8]         # two channels tie for the maximum
    for seed in range(20):
        random.seed(seed)
        ref = tv082.color_jitter_get_params((0.8, 1.2), (0.8, 1.2), (0.8, 1.2), (-0.1, 0.1))
        state = random.getstate()
        random.seed(seed)
        mine = cj.get_params((0.8, 1.2), (0.8, 1.2), (0.8, 1.2), (-0.1, 0.1))
        assert random.getstate() == state               # same number of draws from the same generator
        a, b = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ya, yb = ref(a), mine(b)
        assert float((ya - yb).abs().max()) < 1e-12, (seed, mine)
        w = torch.rand(ya.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(seed))
        (ya * w).sum().backward()
        (yb * w).sum().backward()
        assert float((a.grad - b.grad).abs().max()) < 1e-9, (seed, mine)
    with __import__("pytest").raises(ValueError):
        cj._hue(x, 0.7)



def test_step_log_writes_one_line_per_iteration_one_iteration_late(tmp_path):
    """trainer.StepLog (metrics / logging): phases are the intervals between consecutive marks, named by
    the mark that ends them; the line of iteration i is written at the end of iteration i + 1 (or by close()); no path = off."""
    import json
    import time

    from depthmodelhardening_amd.trainer import StepLog
    path = str(tmp_path / "sub" / "steps.jsonl")
    log = StepLog(path, 32)
    log.cuda = False                 # host clocks: this test has no GPU
    for i in range(3):
        log.mark("start")
        time.sleep(0.004)
        log.mark("attack")
        time.sleep(0.002)
        log.mark("forward+loss")
        log.end_step(i, 0, torch.tensor(1.5 + i))
        assert len(open(path).read().splitlines()) == i         # one iteration late
    log.close()
    lines = [json.loads(l) for l in open(path)]
    assert [l["step"] for l in lines] == [0, 1, 2] and [l["loss"] for l in lines] == [1.5, 2.5, 3.5]
    for l in lines:
        assert set(l["phase_ms"]) - {"between_steps"} == {"attack", "forward+loss"}
        assert l["phase_ms"]["attack"] >= 3.9 > l["phase_ms"]["forward+loss"] >= 1.9
        # the headline rate is the device time of the loop body, not the host's enqueue interval
        assert abs(l["images_per_s"] * l["device_ms"] / 32e3 - 1) < 1e-2 and 5.9 <= l["device_ms"] < 1000
    assert "between_steps" not in lines[0]["phase_ms"] and all("between_steps" in l["phase_ms"] for l in lines[1:])
    assert "host_enqueue_ms" not in lines[0] and all(6 <= l["host_enqueue_ms"] < 1000 for l in lines[1:])
    off = StepLog("", 32)
    off.mark("start")
    off.end_step(0, 0, 1.0)
    off.close()
    assert not off.marks and off.file is None
