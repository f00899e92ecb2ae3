"""Length-aware vocoder mode and the batched long-form path.

The reference's TextToSpeech::call (/root/reference/cpp/helper.cpp:685-722) runs one batch-of-one _infer per text chunk.
The engine runs the chunks as ONE batch; for that to keep the reference's semantics, row b of the batch must equal what
utterance b gives on its own.  The masked stages (duration, text encoder, vector estimator) have that property by
construction; the vocoder gets it from `stn_set_vocoder_mode(h, 1)`.  Checked here against the CPU oracle run one
utterance at a time, and against the engine's own batch-of-one runs at the full architecture in both dtypes."""
import numpy as np
import pytest

from oracle.neural_ref import RefModel, randn
from supertonic_amd import binding
from supertonic_amd.arch import default_arch, tiny_arch
from gpu_util import make_inputs, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("C,k,dil,B,L", [(64, 3, 1, 3, 37),      # generic kernel (k=3)
                                          (384, 5, 2, 3, 37),     # 2 frames per wave
                                          (512, 7, 4, 9, 471),    # comb R=4
                                          (512, 7, 2, 70, 470),   # comb R=8
                                          (256, 5, 8, 200, 94)])
@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_dwconv_ln_ragged_equals_per_sequence(C, k, dil, B, L, dtype):
    eng = binding.Engine(0, "f32")
    rng = np.random.default_rng(C + k + dil + B)
    x = rng.standard_normal((B, L, C)).astype(np.float32)
    w = rng.standard_normal((C, k)).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    bt = (0.1 * rng.standard_normal(C)).astype(np.float32)
    lens = rng.integers(1, L + 1, B).astype(np.int32)
    lens[0] = L
    lens[-1] = 1
    if B > 2:
        lens[1] = 0  # an empty sequence: all of its rows come back as zeros
    got = eng.op_dwconv_ln(x, w, b, g, bt, dil, dtype=dtype, seqlen=lens)
    for i in range(B):
        n = int(lens[i])
        assert np.all(got[i, n:] == 0.0)
        if n == 0:
            continue
        alone = eng.op_dwconv_ln(np.ascontiguousarray(x[i:i + 1, :n]), w, b, g, bt, dil, dtype=dtype)
        # same arithmetic order in every kernel variant (bias, taps in order, wave reduction): bit-identical
        np.testing.assert_array_equal(got[i, :n], alone[0])


def test_length_aware_batch_matches_oracle_one_by_one():
    a = tiny_arch()
    ref = RefModel(a, 7)
    eng = binding.Engine(0, "f32")
    eng.load_synthetic(a, 7)
    B, Lt = 3, 19
    lens = np.array([19, 7, 12])
    ids, mask, sttl, sdp = make_inputs(a, B, Lt, lens, seed=3)
    durs = np.array([1.3, 0.31, 0.74], np.float32)
    steps, speed = 3, 1.05
    D = a.latent_channels
    cs = a.base_chunk_size * a.chunk_compress_factor
    # geometry of the batch
    eng.set_vocoder_mode(True)
    eng.batch_upload(ids, mask, sttl, sdp, durs)
    eng.batch_run(steps, speed, 1)
    _, L, W = eng.batch_dims()
    noise = randn(11, B, D, L)
    w, d = eng.synthesize(ids, mask, sttl, sdp, steps, speed, noise=noise, duration_override=durs)
    eng.set_vocoder_mode(False)
    w_pad, _ = eng.synthesize(ids, mask, sttl, sdp, steps, speed, noise=noise, duration_override=durs)
    differs = 0
    for i in range(B):
        n = int(lens[i])
        rw, rd = ref.synthesize(ids[i:i + 1, :n], mask[i:i + 1, :, :n], sttl[i:i + 1], sdp[i:i + 1], steps, speed,
                                lambda b_, d_, l_: np.ascontiguousarray(noise[i:i + 1, :, :l_]),
                                duration_override=durs[i:i + 1])
        m = rw.shape[1]
        assert m <= W and m % cs == 0
        np.testing.assert_allclose(d[i], rd[0], rtol=1e-6)
        mx, _ = rel_err(w[i, :m], rw[0])
        assert mx < 2e-3, (i, mx)
        assert np.all(w[i, m:] == 0.0)
        if m < W:
            differs += rel_err(w_pad[i, :m], rw[0])[0] > 1e-3
    # the reference's padded vocoder does NOT have this property (zero latent is signal to the convolutions)
    assert differs >= 1


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-4), ("bf16", 3e-2), ("f16", 4e-3)])
def test_length_aware_batch_matches_batch_of_one_full_arch(dtype, tol):
    a = default_arch()
    eng = binding.Engine(0, dtype)
    eng.load_synthetic(a, 1)
    B, Lt = 4, 64
    lens = np.array([64, 20, 41, 9])
    ids, mask, sttl, sdp = make_inputs(a, B, Lt, lens, seed=5)
    durs = np.array([2.1, 0.7, 1.4, 0.35], np.float32)
    utt = np.array([10, 11, 12, 13], np.int64)
    cs = a.base_chunk_size * a.chunk_compress_factor
    eng.set_vocoder_mode(True)
    w, d = eng.synthesize(ids, mask, sttl, sdp, 5, 1.05, duration_override=durs, noise_seed=99, utt_ids=utt)
    _, L, W = eng.batch_dims()
    for i in range(B):
        n = int(lens[i])
        wi, di = eng.synthesize(ids[i:i + 1, :n], mask[i:i + 1, :, :n], sttl[i:i + 1], sdp[i:i + 1], 5, 1.05,
                                duration_override=durs[i:i + 1], noise_seed=99, utt_ids=utt[i:i + 1])
        m = wi.shape[1]
        assert m % cs == 0 and m <= W
        assert d[i] == di[0]
        mx, rms = rel_err(w[i, :m], wi[0])
        assert mx < tol, (dtype, i, mx, rms)
        assert np.all(w[i, m:] == 0.0)
    eng.set_vocoder_mode(False)


def test_mode_switch_invalidates_graph():
    """A captured graph of one mode must not be replayed for the other."""
    a = tiny_arch()
    eng = binding.Engine(0, "f32")
    eng.load_synthetic(a, 7)
    ids, mask, sttl, sdp = make_inputs(a, 2, 11, np.array([11, 4]), seed=1)
    durs = np.array([0.9, 0.3], np.float32)
    outs = {}
    for mode in (False, True, False, True):
        eng.set_vocoder_mode(mode)
        for _ in range(3):  # eager, capture, replay
            w, _ = eng.synthesize(ids, mask, sttl, sdp, 2, 1.0, duration_override=durs, noise_seed=5)
            if mode in outs:
                np.testing.assert_array_equal(w, outs[mode])
            outs[mode] = w
    assert eng.graph_replays >= 4
    assert not np.array_equal(outs[False], outs[True])


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_graph_replay_survives_shape_changes(dtype):
    """Regression: a captured graph holds raw device pointers.  The batch buffers used to be re-allocated by every upload,
    so after a request of another shape the allocator could hand the SAME xt/wav blocks back with the small buffers
    (utterance ids, lengths) permuted — the replay then drew different noise.  Buffers are now persistent and the graph
    key carries the allocation generation; every replay must reproduce the eager result bit for bit."""
    a = tiny_arch()
    eng = binding.Engine(0, dtype)
    eng.load_synthetic(a, 7)
    shapes = {"A": (2, 11, [11, 4], [0.9, 0.3]), "B": (3, 9, [9, 5, 2], [0.5, 0.4, 0.2]), "C": (2, 11, [11, 4], [0.7, 0.3]),
              "D": (5, 17, [17, 3, 9, 1, 12], [1.4, 0.3, 0.8, 0.2, 1.0])}
    ins = {k: make_inputs(a, B, Lt, np.array(lens), seed=1) + (np.array(d, np.float32),) for k, (B, Lt, lens, d) in shapes.items()}
    want = {}
    for k in "ABCDDDBBBAAACCCDBDBAB":
        ids, mask, sttl, sdp, durs = ins[k]
        w, _ = eng.synthesize(ids, mask, sttl, sdp, 2, 1.0, duration_override=durs, noise_seed=5)
        if k in want:
            np.testing.assert_array_equal(w, want[k], err_msg=k)
        else:
            want[k] = w
    assert eng.graph_replays >= 6
