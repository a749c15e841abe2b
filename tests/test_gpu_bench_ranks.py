"""bench.py's N > 1 branch on the ONE GPU this suite has: two ranks share GPU 0 over gloo (RCCL needs a GPU per rank; the
driver runs the real 8-GPU bench).  What runs here is everything of that branch but the ring itself: the launcher, the
rank-0-first warm-up, shorten_timeout, the timed region with the side-stream all-reduce overlapped with the next attack, the
second timing in the other attack / all-reduce order, the all-reduce's HIP-event durations in the JSON line, and a rank that
stops taking part.  The ranks are started by the launcher BEFORE anything touches the GPU (bench.launch_ranks)."""
import json
import os
import subprocess
import sys
import time

import pytest

# one xdist group: these tests start rank processes of their own, and the box allows six processes on the GPU
pytestmark = [pytest.mark.gpu, pytest.mark.xdist_group("ranks")]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch_size", "4", "--atk_scenes", "2", "--no_cpu_baseline"]


def _bench(extra=(), env=None, timeout=600):
    e = dict(os.environ, DMH_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "DMH_BENCH_CHILD"):
        e.pop(k, None)
    e.update(env or {})
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + FLAGS + list(extra), cwd=REPO, env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    return r, time.time() - t0


def _line(r):
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-4000:])
    return json.loads(lines[0])


@pytest.mark.parametrize("shared", [False, True])
def test_two_ranks_on_one_gpu(shared):
    r, _ = _bench(["--shared_patch"] if shared else [])
    assert r.returncode == 0, r.stderr[-4000:]
    assert "rank 0 warmed the kernel cache" in r.stderr
    d = _line(r)
    cfg = d["config"]
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert cfg["global_batch"] == 8 and cfg["parallelism"] == "dp2" and cfg["attack_overlap"] is True
    assert cfg["shared_patch"] is shared and cfg["attack_scenes"] == 2
    assert cfg["final_loss"] == cfg["final_loss"] and abs(cfg["final_loss"]) < 1e3          # finite
    other = cfg["other_order"]
    assert other["mode"] == "sync_attack" and other["value"] > 0 and other["ms_per_step"] > 0
    assert abs(d["value"] * d["ms_per_step"] / 8e3 - 1) < 1e-3
    # the gradient exchange of every timed step, by HIP events on the side stream
    ar = cfg["all_reduce"]
    assert ar["launches"] == 2 and ar["bucket_mb"] == pytest.approx(57.3, abs=0.05) and ar["backend"] == "gloo"
    assert 0 < ar["all_reduce_ms_per_step"] <= ar["all_reduce_ms_max"] * (1 + 1e-6)
    assert 0 <= ar["optimizer_wait_ms_per_step"] <= ar["optimizer_wait_ms_max"] + 1e-6
    assert d["roofline"]["kernel"] == "photo_bwd_kernel" and "cpu_baseline" not in d


def test_a_rank_that_stops_fails_the_job_within_the_shortened_timeout():
    """Rank 1 stops taking part after the warm-up: rank 0 sits in the next step's gradient all-reduce until the process group's
    steady-state timeout (DMH_DIST_STEADY_TIMEOUT_MIN, here 0.1 min) fails it; the launcher then stops rank 1 and exits
    non-zero -- minutes earlier than the 60-minute start-up timeout would."""
    r, took = _bench(env={"DMH_BENCH_AFTER_WARMUP": "hang:1", "DMH_DIST_STEADY_TIMEOUT_MIN": "0.1"}, timeout=420)
    assert r.returncode != 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "stopping the other ranks" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert took < 400


def test_a_rank_that_dies_fails_the_job_at_once():
    r, _ = _bench(env={"DMH_BENCH_AFTER_WARMUP": "exit:1"}, timeout=420)
    assert r.returncode != 0 and "rank 1 exited with 3" in r.stderr
