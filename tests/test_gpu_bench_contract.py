"""bench.py's one JSON line carries every field the driver and the judge read (task contract, measurement section)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_line_contract():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-sample", "2"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["n_gpus"] == int(os.environ.get("WORLD_SIZE", "1"))
    # the timed region is the production path: hipGraph replays of the post-duration pipeline, exactly K of them; the event-timed
    # (eager) pass that feeds `roofline` is a separate K steps printed beside it
    assert d["graph_replays_in_timed_region"] == d["steps"] and d["config"]["warmup_steps_run"] >= 3
    assert d["eager_sampled"]["ms_per_step"] > 0 and d["eager_sampled"]["ms_per_step"] > 0.9 * d["ms_per_step"]
    assert d["vs_baseline"] is None and d["unit"] == "audio-sec/sec" and d["dtype"] == "bf16" and "synthetic" in d["data"]
    # one GPU needs no PyTorch: the run stays on the HIP runtime the library was built against, and says so; no version-mismatch warning
    assert d["config"]["torch_in_process"] is False and d["config"]["hip_runtime"] // 100000 == d["config"]["hip_built"] // 100000, d["config"]
    assert "was built against HIP" not in p.stderr, p.stderr[-1500:]
    assert d["warmup_run"] >= 3
    # a lone batch on the critical path of a predicted-duration run (predictor -> host read -> rest) costs at least the forced-duration one
    assert d["lone_batch_predicted_path"]["p50_ms"] >= 0.97 * d["p50_latency_ms"]
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 1000 and abs(d["value"] - d["config"]["audio_sec_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0 < r["frac"] < 1 and r["avg_launch_us"] > 0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"]
    assert d["value"] / c["value"] > 50  # a GPU engine that is not far ahead of 16 host threads is broken
    assert "1 warm-up + 3 timed" in c["sample"]
    # host-to-host rate beside the device-resident one (the contract of _infer's return, cpp/helper.cpp:674-682)
    for k in ("value_host", "ms_per_step_host", "p50_latency_host_ms", "host_loop"):
        assert k in d, k
    # the host loop keeps two batches in flight (two engine handles), `value` runs one at a time: the host-to-host rate is bounded
    # by the resident two-in-flight rate, not by `value`
    t2 = d["two_in_flight"]
    assert d["config"]["in_flight_batches"] == 1 and t2["value"] > 0.9 * d["value"]
    assert 0 < d["value_host"] <= t2["value"] * 1.03 and d["ms_per_step_host"] >= t2["ms_per_step"] * 0.97
    assert abs(d["value_host"] - d["config"]["audio_sec_per_step"] / (d["ms_per_step_host"] * 1e-3)) / d["value_host"] < 1e-3
    pc = d["host_loop"]["pcie_bytes_per_step"]
    assert pc["d2h"] > 10e6 and pc["h2d"] > 1e6 and d["p50_latency_host_ms"] >= d["ms_per_step"] * 0.98
    assert "forced" in d["config"]["durations"]
    su = d["single_utterance"]
    assert su["f32"]["p50_ms"] > 0 and su["bf16"]["p50_ms"] > 0 and su["bf16"]["p50_ms"] < su["f32"]["p50_ms"]
    if r.get("profile_stale"):
        assert r["traffic"] is None


def test_bench_strong_scaling_flag():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--scaling", "strong",
                        "--no-host-loop", "--no-b1", "--no-profile"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert d["scaling"] == "strong" and d["config"]["global_batch"] == 128 and d["config"]["batch_per_gpu"] == 128


def test_bench_multi_gpu_code_path_at_world_1():
    """STN_BENCH_FORCE_DIST=1: the N > 1 code path on the one GPU a test box has — `nccl` process group (RCCL), the engine on the shared
    torch stream, GatherPlan, the int16 PCM gather overlapped with the next step — and what rank 0 gathered equals stn_batch_fetch_pcm16."""
    env = dict(os.environ, STN_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--no-host-loop", "--no-b1",
                        "--no-profile"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    g = d["gather_check"]
    assert g["equal_rank0_block"] and g["durations_equal"] and g["crc32_gathered"] == g["crc32_fetch_pcm16"]
    assert g["blocks"] == [[128, g["blocks"][0][1]]] and all(g["nonzero_blocks"]) and g["bytes_per_gather"] > 10e6
    assert d["n_gpus"] == 1 and d["graph_replays_in_timed_region"] == 2 and d["value"] > 1000


def test_bench_refuses_more_gpus_than_the_box_has():
    from supertonic_amd import binding
    n = binding.device_count()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "1"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    assert p.returncode != 0 and "visible" in p.stderr and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
