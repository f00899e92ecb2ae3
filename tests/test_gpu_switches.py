"""Every diagnostic switch of the engine (environment variables read at stn_create / at a kernel's first launch, honoured only beside the master
switch STN_DEV_SWITCHES=1: csrc/dev_env.hpp) is a configuration somebody will run: each one is driven here through a whole synthesis of the default (full-width) descriptor in a fresh process and
held against the default configuration — bit-equal where the switch only moves work between streams, within the dtype's recorded bound where it
picks another kernel form (stn.h, "WHAT IS AND IS NOT BIT-IDENTICAL").  A malformed value must not take the process down either."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys, hashlib
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from supertonic_amd import binding
from supertonic_amd.arch import default_arch
from gpu_util import make_inputs
a = default_arch()
out = {}
for dtype in ("bf16", "f32"):
    eng = binding.Engine(0, dtype)
    eng.load_synthetic(a, 11)
    B, Lt = 24, 40
    lens = np.array([Lt] + [5 + (7 * i) % (Lt - 5) for i in range(B - 1)])
    ids, mask, sttl, sdp = make_inputs(a, B, Lt, lens, seed=4)
    durs = np.linspace(0.4, 2.6, B).astype(np.float32)
    w, d = eng.synthesize(ids, mask, sttl, sdp, 3, 1.05, duration_override=durs, noise_seed=5)
    w2, d2 = eng.synthesize(ids, mask, sttl, sdp, 3, 1.05, duration_override=durs, noise_seed=5)   # the replayed graph
    assert np.array_equal(w, w2) and np.array_equal(d, d2)
    # the predictor's own lengths too (no override)
    _, dp = eng.synthesize(ids, mask, sttl, sdp, 1, 1.05, noise_seed=5)
    np.save(sys.argv[2] + "_" + dtype + ".npy", w)
    out[dtype] = {"sha": hashlib.sha256(w.tobytes()).hexdigest(), "dur": [float(x) for x in dp], "rows": int(eng.ve_rows)}
print("RESULT " + json.dumps(out))
"""

# switch -> (value, "equal" | "bound")
SWITCHES = [
    ("STN_DP_STREAM", "0", "equal"),       # duration predictor / text encoder on the main stream instead of beside it
    ("STN_PACKED", "0", "bound"),          # padded rows: other launch shapes, so other kernel forms (K4-split threshold)
    ("STN_NT", "0", "equal"),              # streaming-store hints off
    ("STN_FFN", "0", "bound"),             # every pointwise pair as two GEMM launches
    ("STN_FFN", "1", "bound"),             # K4 without the hidden split
    ("STN_FFN_SPLIT_S", "4", "bound"),     # the 4-way split at every size
    ("STN_FFN_SPLIT_S", "8", "bound"),     # the 8-way split at every size
    ("STN_FFN_SPLIT_MIN_ROWS", "100000", "bound"),
    ("STN_FFN_MIN_ROWS", "1", "bound"),    # vocoder K4 at every size
    ("STN_GEMM_TR", "0", "bound"),
    ("STN_XATTN", "0", "bound"),           # the cross-attention blocks as four launches instead of head-split
    ("STN_FOLD_TCH", "8", "equal"),        # the fold kernel's run length: the same bits either way
    ("STN_FOLD_TCH", "32", "equal"),
    ("STN_FOLD_TCH", "40", "equal"),       # (two passes of the conv / LayerNorm phase for runs longer than 32 frames)
    ("STN_FOLD_TCH", "48", "equal"),
    ("STN_DWCONV_XCD", "0", "equal"),      # tile order of the comb kernel: placement only
    ("STN_PRIO", "hn", "equal"),           # stream priorities: scheduling only
    ("STN_PRIO", "ll", "equal"),
    ("STN_FFN", "banana", "bound"),        # malformed: falls back to a documented value, never crashes
    ("STN_FFN_SPLIT_S", "5", "bound"),     # not a supported split: ignored
]


def _run(tmp, tag, env_extra, master=True):
    env = dict(os.environ)
    for k in [s[0] for s in SWITCHES] + ["STN_DEV_SWITCHES"]:
        env.pop(k, None)
    env.update(env_extra)
    if env_extra and master:
        env["STN_DEV_SWITCHES"] = "1"
    prefix = os.path.join(tmp, tag)
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, prefix], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (tag, r.stdout[-2000:], r.stderr[-4000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    res = json.loads(line[7:])
    for dt in res:
        res[dt]["wav"] = np.load(prefix + "_" + dt + ".npy")
    res["stderr"] = r.stderr
    return res


def test_every_environment_switch_against_the_default(tmp_path):
    tmp = str(tmp_path)
    base = _run(tmp, "default", {})
    again = _run(tmp, "default2", {})
    dts = [k for k in base if k != "stderr"]
    for dt in dts:
        assert base[dt]["sha"] == again[dt]["sha"], dt   # process-to-process determinism of the default
    # a switch without the master is named on stderr and changes nothing: the default configuration's bits and launch shapes
    stray = _run(tmp, "stray", {"STN_FFN": "0", "STN_XATTN": "0", "STN_PACKED": "0"}, master=False)
    for dt in dts:
        assert stray[dt]["sha"] == base[dt]["sha"] and stray[dt]["rows"] == base[dt]["rows"], dt
    for name in ("STN_FFN", "STN_XATTN", "STN_PACKED"):
        assert stray["stderr"].count(name + "=0 ignored") == 1, stray["stderr"][-2000:]
    assert "ignored" not in base["stderr"]
    bounds = {"bf16": 6e-2, "f32": 5e-5}
    for i, (name, val, kind) in enumerate(SWITCHES):
        got = _run(tmp, "s%d" % i, {name: val})
        assert "ignored" not in got["stderr"]
        for dt in dts:
            # the predictor runs fp32 in every engine: its durations are bit-equal under the switches that only move work, and equal to fp32
            # rounding under those that change a launch's row count or tile (padded text rows leave the M <= 512 split-K regime of the fp32 GEMM:
            # another summation order, stn.h "WHAT IS AND IS NOT BIT-IDENTICAL")
            if kind == "equal":
                assert got[dt]["dur"] == base[dt]["dur"], (name, val, dt)
            else:
                np.testing.assert_allclose(got[dt]["dur"], base[dt]["dur"], rtol=2e-5, err_msg=str((name, val, dt)))
            w0, w1 = base[dt]["wav"], got[dt]["wav"]
            assert w0.shape == w1.shape
            if kind == "equal":
                assert got[dt]["sha"] == base[dt]["sha"], (name, val, dt)
            else:
                scale = float(np.max(np.abs(w0))) + 1e-12
                mx = float(np.max(np.abs(w0.astype(np.float64) - w1))) / scale
                assert mx < bounds[dt], (name, val, dt, mx)


C3_CHILD = r"""
import sys, hashlib
import numpy as np
sys.path.insert(0, sys.argv[1])
from supertonic_amd import binding, host, workload
from supertonic_amd.arch import default_arch
a = default_arch()
n = 128
texts = workload.utterances(n, 10, seed=1234)
ids, mask = host.UnicodeProcessor(host.synthetic_indexer())(texts, ["en"] * n)
sttl, sdp = workload.synthetic_styles(a, np.arange(n))
eng = binding.Engine(0, "bf16")
eng.load_synthetic(a, 7)
w, d = eng.synthesize(ids, mask, sttl, sdp, 2, 1.05, duration_override=workload.forced_durations(texts), noise_seed=5)
lat = (np.ceil(d * a.sample_rate / (a.base_chunk_size * a.chunk_compress_factor))).astype(np.int32)
print("RESULT", hashlib.sha256(w.tobytes()).hexdigest(), binding.fold_run_frames(lat, 256), int(lat.max()))
"""


def test_the_bench_shape_is_bit_identical_under_every_fold_run_length(tmp_path):
    """At the bench's shape (128 ten-word utterances) the fold kernel's run length is chosen from the lengths (40 frames: one round of workgroups,
    tests/test_fold_run_cpu.py); the waveform must not depend on it: the default against runs of 32 and of 48 frames forced."""
    out = {}
    for tag, extra in (("auto", {}), ("32", {"STN_DEV_SWITCHES": "1", "STN_FOLD_TCH": "32"}), ("48", {"STN_DEV_SWITCHES": "1", "STN_FOLD_TCH": "48"})):
        env = dict(os.environ)
        for k in ("STN_DEV_SWITCHES", "STN_FOLD_TCH"):
            env.pop(k, None)
        env.update(extra)
        r = subprocess.run([sys.executable, "-c", C3_CHILD, ROOT], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (tag, r.stdout[-2000:], r.stderr[-4000:])
        out[tag] = [l.split() for l in r.stdout.splitlines() if l.startswith("RESULT")][-1]
    assert out["auto"][2] == "40" and int(out["auto"][3]) > 64          # (the choice this test is about)
    assert out["auto"][1] == out["32"][1] == out["48"][1]
