"""stn_batch_run keeps its two text stages (duration predictor, text encoder) on side streams, beside each other and beside the tail
of the previous run; the latent pipeline waits for the encoder's rows through an event and a copy.  Whatever the call pattern —
runs queued back to back without a sync, batches of other shapes in between, predicted or forced durations — every run must return
exactly what an engine that does everything on one stream returns (STN_DP_STREAM=0).  Order of the stages is not observable in
the reference (`_infer`, /root/reference/cpp/helper.cpp:512-556 runs them one after the other); only the results are."""
import os

import numpy as np
import pytest

from supertonic_amd import binding
from supertonic_amd.arch import default_arch, tiny_arch
from gpu_util import make_inputs

pytestmark = pytest.mark.gpu


def _single_stream_engine(arch, dtype):
    os.environ["STN_DP_STREAM"] = "0"
    os.environ["STN_DEV_SWITCHES"] = "1"  # (the master switch the measurement switches need: csrc/dev_env.hpp)
    try:
        e = binding.Engine(0, dtype)
    finally:
        del os.environ["STN_DP_STREAM"], os.environ["STN_DEV_SWITCHES"]
    e.load_synthetic(arch, 7)
    return e


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_back_to_back_runs_and_shape_changes_equal_the_single_stream_engine(dtype):
    a = tiny_arch()
    eng = binding.Engine(0, dtype)
    eng.load_synthetic(a, 7)
    ref = _single_stream_engine(a, dtype)
    rng = np.random.default_rng(5)
    batches = []
    for case in range(6):
        B, Lt = int(rng.integers(1, 7)), int(rng.integers(6, 20))
        lens = rng.integers(2, Lt + 1, B)
        lens[0] = Lt
        batches.append(make_inputs(a, B, Lt, lens, seed=100 + case) + (rng.uniform(0.15, 0.9, B).astype(np.float32),))
    for rnd in range(3):
        for case, i in enumerate(batches):
            forced = (case + rnd) % 2 == 0
            kw = dict(duration_override=i[4]) if forced else {}
            want = None
            for e in (ref, eng):
                e.batch_upload(*i[:4], **kw)
                for _ in range(1 + (case % 3)):  # queued back to back, no sync in between
                    e.batch_run(2, 1.0, 7 + case)
                got = e.batch_fetch()
                if e is ref:
                    want = got
            np.testing.assert_array_equal(got[0], want[0], err_msg=f"round {rnd} case {case} forced={forced}")
            np.testing.assert_array_equal(got[1], want[1])


def test_full_size_back_to_back_is_bit_stable():
    """The bench's own pattern at the bench's size: the same resident C3-sized batch run 6 times without a sync, graph replays
    included, against the single-stream engine."""
    from supertonic_amd import host, workload
    a = default_arch()
    texts = workload.utterances(128, 10, seed=1234)
    ids, mask = host.UnicodeProcessor(host.synthetic_indexer())(texts, ["en"] * 128)
    sttl, sdp = workload.synthetic_styles(a, list(range(128)))
    durs = workload.forced_durations(texts)
    outs = []
    for make in (lambda: _single_stream_engine(a, "bf16"), lambda: (lambda e: (e.load_synthetic(a, 7), e)[1])(binding.Engine(0, "bf16"))):
        e = make()
        e.batch_upload(ids, mask, sttl, sdp, duration_override=durs)
        for _ in range(6):
            e.batch_run(5, 1.05, 1234)
        outs.append(e.batch_fetch()[0])
        e.batch_run(5, 1.05, 1234)
        np.testing.assert_array_equal(e.batch_fetch()[0], outs[-1])
    np.testing.assert_array_equal(outs[0], outs[1])
