"""`python bench.py --gpus N` with no launcher around it starts its N ranks itself (the driver's command form
for N = 1 is exactly this).  CPU: the spawn / rendezvous / max-over-ranks / one-JSON-line skeleton runs with a stand-in step over
gloo (STN_BENCH_STUB=1 — nothing it prints is a measurement), and without GPUs the real path refuses instead of running fewer ranks."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=300):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


def test_gpus_2_spawns_two_ranks_and_prints_one_line():
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8"], {"STN_BENCH_STUB": "1"})
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 16 and d["config"]["batch_per_gpu"] == 8
    assert "starting 2 ranks" in p.stderr
    # the max over ranks: the stand-in step of rank 1 sleeps twice as long as rank 0's
    assert d["ms_per_step"] >= 3.5


def test_strong_scaling_deals_one_batch_over_the_ranks():
    p = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8", "--scaling", "strong"], {"STN_BENCH_STUB": "1"})
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["global_batch"] == 8 and d["config"]["batch_per_gpu"] == 4


def test_more_gpus_than_visible_fails_loudly():
    """No stub: this container has no GPU, a GPU box has one — `--gpus 2` must exit non-zero, not run one rank under that label."""
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "1"], {"STN_BENCH_STUB": "0", "HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    assert p.returncode != 0
    assert "only" in p.stderr and "visible" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_launcher_and_flag_must_agree():
    """WORLD_SIZE set by a launcher but different from --gpus: an error (before any GPU work), not a silent choice of one of them."""
    p = _run(["--gpus", "4", "--steps", "1", "--warmup", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "STN_BENCH_STUB": "0"})
    assert p.returncode != 0 and "disagree" in p.stderr
