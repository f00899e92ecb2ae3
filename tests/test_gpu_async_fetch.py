"""The pipelined waveform fetch (stn_batch_fetch_pcm16_begin/_end: PCM conversion + device->host copy on a second stream into the
handle's pinned slots) and the measurement hooks added with it (launch log, sampled event timing), on a real MI355X.
The PCM is what writeWavFile stores (/root/reference/cpp/helper.cpp:986-987); the synchronous stn_batch_fetch_pcm16 is the checker."""
import numpy as np
import pytest

from supertonic_amd import binding
from supertonic_amd.arch import tiny_arch
from gpu_util import make_inputs

pytestmark = pytest.mark.gpu


def _batch(a, B, Lt, seed):
    lens = np.linspace(Lt, 3, B).astype(int)
    return make_inputs(a, B, Lt, lens, seed=seed) + (np.linspace(0.9, 0.2, B).astype(np.float32),)


def test_async_fetch_equals_the_synchronous_one_and_slots_keep_their_batch():
    a = tiny_arch()
    eng = binding.Engine(0, "bf16")
    eng.load_synthetic(a, 7)
    with pytest.raises(binding.StnError, match="no fetch in flight"):
        eng.fetch_pcm16_end(0)
    with pytest.raises(binding.StnError):
        eng.fetch_pcm16_begin(2)
    i4, i3 = _batch(a, 4, 12, 1), _batch(a, 3, 9, 2)
    want = {}
    for name, i in (("four", i4), ("three", i3)):
        eng.batch_upload(*i[:4], duration_override=i[4])
        eng.batch_run(2, 1.0, 5)
        want[name] = eng.batch_fetch_pcm16()
    # batch "four" into slot 0, then ANOTHER batch (other B, other length) is uploaded, run and started on slot 1 before slot 0 is read
    eng.batch_upload(*i4[:4], duration_override=i4[4])
    eng.batch_run(2, 1.0, 5)
    eng.fetch_pcm16_begin(0)
    eng.batch_upload(*i3[:4], duration_override=i3[4])
    eng.batch_run(2, 1.0, 5)
    eng.fetch_pcm16_begin(1)
    pcm0, d0 = eng.fetch_pcm16_end(0)
    pcm1, d1 = eng.fetch_pcm16_end(1, copy=False)
    assert pcm0.dtype == np.int16 and pcm0.shape == want["four"][0].shape and pcm1.shape == want["three"][0].shape
    np.testing.assert_array_equal(pcm0, want["four"][0])
    np.testing.assert_array_equal(pcm1, want["three"][0])
    np.testing.assert_array_equal(d0, want["four"][1])
    np.testing.assert_array_equal(d1, want["three"][1])
    # a slot can be reused at once (its previous copy is waited for inside _begin), and ended twice (the buffer stays valid)
    eng.fetch_pcm16_begin(1)
    again, _ = eng.fetch_pcm16_end(1)
    np.testing.assert_array_equal(again, want["three"][0])
    np.testing.assert_array_equal(eng.fetch_pcm16_end(1)[0], again)


def test_launch_log_and_sampled_timing():
    a = tiny_arch()
    eng = binding.Engine(0, "bf16")
    eng.load_synthetic(a, 7)
    i = _batch(a, 3, 10, 3)
    eng.batch_upload(*i[:4], duration_override=i[4])
    eng.batch_run(2, 1.0, 5)  # sizes the workspace
    eng.profile_filter(None)
    eng.profile_enable(True)
    eng.launch_log_enable(True)
    eng.profile_reset()
    eng.batch_run(2, 1.0, 5)
    eng.sync()
    log = eng.launch_log()
    stats = eng.profile()
    tagged = [f for f, _ in log if f != "-"]
    assert len(log) > len(tagged) > 50                                         # glue kernels carry no family
    assert sum(v["launches"] for v in stats.values()) == len(tagged)           # one log entry per timed launch, same order of magnitude
    assert {"dp", "te", "ve", "vo"} == {f.split(".")[0] for f in tagged}
    kernels = {k for _, k in log}
    assert any("gemm" in k for k in kernels) and any("dwconv_ln" in k for k in kernels) and all(" " not in k and "<" not in k for k in kernels)
    fam = max(stats, key=lambda k: stats[k]["launches"])
    n_all = stats[fam]["launches"]
    # every 3rd launch of one family only
    eng.profile_filter(fam)
    eng.profile_sample(3)
    eng.profile_reset()
    eng.batch_run(2, 1.0, 5)
    eng.sync()
    s2 = eng.profile()
    assert list(s2) == [fam] and s2[fam]["launches"] == (n_all + 2) // 3 and s2[fam]["ms"] > 0
    eng.profile_sample(1)
    eng.launch_log_enable(False)
    eng.profile_enable(False)
    eng.profile_reset()
    assert eng.launch_log() == []
