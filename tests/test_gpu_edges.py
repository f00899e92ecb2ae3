"""Edge cases of the path on a real MI355X (the reference has no tests; these are the boundaries its host code handles):
empty text, one-frame latents, maximum-length text, single/50 Euler steps, batch/style mismatches, PCM conversion."""
import numpy as np
import pytest

from oracle import host_ref
from oracle.neural_ref import RefModel, randn
from supertonic_amd import binding, host, workload
from supertonic_amd.arch import tiny_arch
from gpu_util import make_inputs, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref():
    return RefModel(tiny_arch(), 7)


@pytest.fixture(scope="module")
def eng():
    e = binding.Engine(0, "f32")
    e.load_synthetic(tiny_arch(), 7)
    return e


def _run_both(ref, eng, ids, mask, sttl, sdp, steps, speed, durs, tol=2e-3):
    nz = {}

    def nf(B, D, L):
        nz["x"] = randn(5, B, D, L)
        return nz["x"]

    rw, rd = ref.synthesize(ids, mask, sttl, sdp, steps, speed, nf, duration_override=durs)
    w, d = eng.synthesize(ids, mask, sttl, sdp, steps, speed, noise=nz["x"], duration_override=durs)
    assert w.shape == rw.shape
    np.testing.assert_allclose(d, rd, rtol=1e-6)
    mx, _ = rel_err(w, rw)
    assert mx < tol, mx
    return w, d


def test_empty_text_and_ragged_batch(ref, eng):
    """'' -> '<en></en>' (9 tokens, no period added: cpp/helper.cpp:156) next to a long utterance."""
    a = tiny_arch()
    up = host.UnicodeProcessor(host.synthetic_indexer())
    ids, mask = up(["", "A considerably longer sentence than the empty one."], ["en", "en"])
    assert mask.sum(axis=(1, 2)).astype(int).tolist() == [9, 59]
    sttl, sdp = workload.synthetic_styles(a, [0, 1])
    _run_both(ref, eng, ids, mask, sttl, sdp, 2, 1.05, np.array([0.08, 0.9], np.float32))


def test_one_latent_frame_and_single_token(ref, eng):
    a = tiny_arch()
    ids, mask, sttl, sdp = make_inputs(a, 1, 1, [1], seed=2)
    w, d = _run_both(ref, eng, ids, mask, sttl, sdp, 1, 1.0, np.array([0.01], np.float32))  # 441 samples -> L = 1
    assert w.shape == (1, a.chunk_size)


def test_maximum_text_length_and_many_steps(ref, eng):
    """chunkText caps a chunk at 300 characters (cpp/helper.cpp:698) -> ~310 tokens with tags; total_step up to 50 (service.py:36)."""
    a = tiny_arch()
    text = ("abcdefghi " * 30).strip()[:299] + "."
    ids, mask = host.UnicodeProcessor(host.synthetic_indexer())([text], ["en"])
    assert ids.shape[1] == 300 + 9
    sttl, sdp = workload.synthetic_styles(a, [3])
    _run_both(ref, eng, ids, mask, sttl, sdp, 50, 2.0, np.array([1.0], np.float32), tol=1e-2)


def test_style_text_count_mismatch_is_an_error(eng):
    """cpp/helper.cpp:479-481 throws 'Number of texts must match number of style vectors'; at the C ABI the batch size is a
    single argument, so the mismatch can only show up as buffers of the wrong size — the host layers check it."""
    from supertonic_amd import host as H
    with pytest.raises(ValueError):
        H.UnicodeProcessor(H.synthetic_indexer())(["a", "b"], ["en"])


def test_pcm16_fetch_matches_reference_wav_conversion(eng):
    a = tiny_arch()
    ids, mask, sttl, sdp = make_inputs(a, 2, 10, [10, 6], seed=4)
    wav, dur = eng.synthesize(ids, mask, sttl, sdp, 2, 1.0, duration_override=np.array([0.2, 0.1], np.float32))
    wav2 = wav * 25.0  # drive part of the signal into clipping
    pcm, dur2 = eng.batch_fetch_pcm16()
    ref_pcm = np.frombuffer(host_ref.wav_bytes(wav.ravel(), a.sample_rate)[44:], dtype="<i2").reshape(wav.shape)
    assert np.array_equal(pcm, ref_pcm) and np.array_equal(dur, dur2)
    # the host-side encoder (C++) agrees with the oracle on clipped data too
    assert host.wav_bytes(wav2.ravel(), a.sample_rate) == host_ref.wav_bytes(wav2.ravel(), a.sample_rate)
    # device-side copy into a strided buffer (the RCCL gather payload of bench.py / supertonic_amd.dist): same samples
    from hip_util import DeviceBuffer
    B, W = wav.shape
    for stride in (W + 8, W + 3):  # 16-byte aligned rows (vector path) and an odd stride (scalar path)
        t = DeviceBuffer(np.full((B, stride), -7, dtype=np.int16))
        eng.batch_copy_pcm16_device(t.ptr, stride)
        eng.sync()
        got = t.to_host()
        assert np.array_equal(got[:, :W], ref_pcm) and np.all(got[:, W:] == -7)


def test_speed_scales_duration_and_length(eng):
    a = tiny_arch()
    ids, mask, sttl, sdp = make_inputs(a, 1, 8, [8], seed=6)
    _, d1 = eng.synthesize(ids, mask, sttl, sdp, 2, 1.0, noise_seed=3)
    _, L1, _ = eng.batch_dims()
    _, d2 = eng.synthesize(ids, mask, sttl, sdp, 2, 2.0, noise_seed=3)
    _, L2, _ = eng.batch_dims()
    np.testing.assert_allclose(d2 * 2, d1, rtol=1e-6)  # duration /= speed (cpp/helper.cpp:529-531)
    assert L2 <= (L1 + 1) // 2 + 1


def test_graph_replay_equals_eager(eng):
    """The hipGraph replay of a shape must reproduce the eager result bit for bit, also when the per-call data (lengths, seed)
    change between replays of the same shape."""
    a = tiny_arch()
    ids, mask, sttl, sdp = make_inputs(a, 3, 10, [10, 7, 4], seed=8)
    d_a = np.array([0.30, 0.28, 0.11], np.float32)
    d_b = np.array([0.30, 0.10, 0.25], np.float32)  # same L (max unchanged), different per-utterance lengths
    eng.set_graph_mode(False)
    eng.batch_upload(ids, mask, sttl, sdp, d_a)
    eng.batch_run(2, 1.0, 11)
    wa, _ = eng.batch_fetch()
    eng.batch_upload(ids, mask, sttl, sdp, d_b)
    eng.batch_run(2, 1.0, 12)
    wb, _ = eng.batch_fetch()
    eng.set_graph_mode(True)
    r0 = eng.graph_replays
    eng.batch_upload(ids, mask, sttl, sdp, d_a)
    for _ in range(3):  # eager warm-up, capture + replay, replay
        eng.batch_run(2, 1.0, 11)
    g1, _ = eng.batch_fetch()
    assert eng.graph_replays >= r0 + 2 and np.array_equal(g1, wa)
    eng.batch_upload(ids, mask, sttl, sdp, d_b)  # new upload: new device buffers -> re-captured; then replayed with new data
    for _ in range(3):
        eng.batch_run(2, 1.0, 12)
    g2, _ = eng.batch_fetch()
    assert np.array_equal(g2, wb)


def test_contract_violations_are_error_codes_not_aborts():
    """include/stn.h: every function returns STN_OK or a negative code.  A head dim the attention kernels do not support and a
    descriptor wider than the LayerNorm kernels handle come back as STN_ERR_INVALID with a message — the process stays alive
    (the reference throws std::runtime_error in the same situations, /root/reference/cpp/helper.cpp:479-481, 193)."""
    from supertonic_amd.arch import default_arch
    e = binding.Engine(0, "bf16")
    q = np.zeros((1 * 4, 2 * 12), np.float32)
    with pytest.raises(binding.StnError) as ei:
        e.op_attention(q.reshape(1, 4, 24), q.reshape(1, 4, 24), q.reshape(1, 4, 24), H=2)
    assert ei.value.code == -1 and "stn_op_attention" in str(ei.value)
    a = default_arch()
    a.vo_dim = 2048
    with pytest.raises(binding.StnError) as ei:
        e.load_synthetic(a, 7)
    assert ei.value.code == -1 and "vo_dim = 2048" in str(ei.value)
    a = default_arch()
    a.ve_heads = 5  # 384 / 5 is not an integer head dim
    with pytest.raises(binding.StnError) as ei:
        e.load_synthetic(a, 7)
    assert ei.value.code == -1 and "ve_heads" in str(ei.value)
    # a GEMM whose K breaks the 16-byte row contract: the launcher throws, the ABI returns the code
    with pytest.raises(binding.StnError):
        e.op_gemm(np.zeros((8, 12), np.float32), np.zeros((8, 12), np.float32), dtype="bf16")
    # the handle is still usable
    e.load_synthetic(tiny_arch(), 7)
    ids, mask, sttl, sdp = make_inputs(tiny_arch(), 1, 6, [6], seed=1)
    w, _ = e.synthesize(ids, mask, sttl, sdp, 2, 1.0, duration_override=np.array([0.3], np.float32))
    assert np.all(np.isfinite(w))
