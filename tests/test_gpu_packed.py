"""Packed ("ragged") rows in the vector estimator: an exact optimisation of the padded [b*L + t] layout — the masked stages are
row-independent, so the latent after the Euler loop must agree between the two layouts (and with the oracle, which the other
GPU tests check on the default, packed, layout)."""
import numpy as np
import pytest

from supertonic_amd import binding, host
from supertonic_amd.arch import default_arch, tiny_arch
from gpu_util import make_inputs, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-5), ("bf16", 2e-2), ("f16", 2.5e-3)])
@pytest.mark.parametrize("arch_fn", [tiny_arch, default_arch])
def test_packed_rows_equal_padded_rows(dtype, tol, arch_fn):
    a = arch_fn()
    eng = binding.Engine(0, dtype)
    eng.load_synthetic(a, 7)
    rng = np.random.default_rng(5)
    for case in range(6):
        B = [1, 2, 5, 9, 3, 16][case]
        Lt = int(rng.integers(4, 50))
        lens = rng.integers(1, Lt + 1, B)
        lens[0] = Lt
        ids, mask, sttl, sdp = make_inputs(a, B, Lt, lens, seed=case)
        durs = rng.uniform(0.08, 2.5 if case else 0.3, B).astype(np.float32)
        if case == 4:
            durs[:] = durs[0]  # no padding at all: both layouts coincide
        steps = [1, 2, 5, 3, 2, 4][case]
        outs = {}
        for packed in (False, True):
            eng.set_packed_rows(packed)
            w, d = eng.synthesize(ids, mask, sttl, sdp, steps, 1.05, duration_override=durs, noise_seed=7 + case)
            outs[packed] = (w, d, eng.batch_fetch_latent())
            _, L, _ = eng.batch_dims()
            lens_lat = host.latent_geometry(d, a.sample_rate, a.base_chunk_size, a.chunk_compress_factor, a.latent_dim)[2]
            supported = a.ve_kernel in (5, 7) and a.ve_dim <= 512  # the comb dwconv kernel carries the packed layout
            assert eng.ve_rows == (int(np.sum(lens_lat)) if packed and supported else B * L), (case, packed, eng.ve_rows)
        np.testing.assert_array_equal(outs[True][1], outs[False][1])
        mx, _ = rel_err(outs[True][2], outs[False][2])
        assert mx < tol, (case, "latent", mx)
        mx, _ = rel_err(outs[True][0], outs[False][0])
        assert mx < 5 * tol, (case, "wav", mx)
        # the padding of the latent is exactly zero in both
        assert np.array_equal(outs[True][2] == 0, outs[False][2] == 0) or dtype in ("bf16", "f16")
    eng.set_packed_rows(True)


def test_packed_graph_replay_and_length_changes():
    """Replays stay exact; a change of the lengths (same shapes otherwise) re-captures instead of replaying a stale grid."""
    a = tiny_arch()
    eng = binding.Engine(0, "f32")
    eng.load_synthetic(a, 7)
    ids, mask, sttl, sdp = make_inputs(a, 4, 12, np.array([12, 5, 9, 2]), seed=2)
    d1 = np.array([1.0, 0.4, 0.7, 0.2], np.float32)
    d2 = np.array([1.0, 0.9, 0.3, 0.6], np.float32)  # same L (max), different sum of lengths
    want = {}
    for k, d in [(1, d1), (1, d1), (1, d1), (2, d2), (2, d2), (2, d2), (1, d1), (2, d2)]:
        w, _ = eng.synthesize(ids, mask, sttl, sdp, 2, 1.0, duration_override=d, noise_seed=3)
        if k in want:
            np.testing.assert_array_equal(w, want[k])
        else:
            want[k] = w
    assert eng.graph_replays >= 2
    eng.set_packed_rows(False)
    w, _ = eng.synthesize(ids, mask, sttl, sdp, 2, 1.0, duration_override=d2, noise_seed=3)
    assert rel_err(w, want[2])[0] < 1e-4


def test_length_aware_vocoder_packed_equals_padded():
    """In the length-aware mode the vocoder runs on packed rows too (bf16): same waveform as on padded rows, exact zeros past
    every utterance's own length."""
    a = default_arch()
    eng = binding.Engine(0, "bf16")
    eng.load_synthetic(a, 7)
    rng = np.random.default_rng(11)
    B, Lt = 7, 40
    lens = rng.integers(3, Lt + 1, B)
    lens[2] = Lt
    ids, mask, sttl, sdp = make_inputs(a, B, Lt, lens, seed=9)
    durs = rng.uniform(0.1, 2.0, B).astype(np.float32)
    eng.set_vocoder_mode(True)
    outs = {}
    for packed in (False, True):
        eng.set_packed_rows(packed)
        for _ in range(3):  # eager, capture, replay
            w, d = eng.synthesize(ids, mask, sttl, sdp, 3, 1.0, duration_override=durs, noise_seed=21)
            if packed in outs:
                np.testing.assert_array_equal(w, outs[packed])
            outs[packed] = w
    mx, rms = rel_err(outs[True], outs[False])
    assert mx < 5e-2 and rms < 5e-3, (mx, rms)
    assert np.array_equal(outs[True] == 0, outs[False] == 0)
    eng.set_vocoder_mode(False)
    eng.set_packed_rows(True)


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_zero_length_utterance_in_a_packed_batch(dtype):
    """A duration of zero gives an utterance no latent frame at all: it owns no row in the packed layout; its waveform is the
    vocoder's response to an all-zero latent, as on padded rows."""
    a = tiny_arch()
    eng = binding.Engine(0, dtype)
    eng.load_synthetic(a, 7)
    ids, mask, sttl, sdp = make_inputs(a, 4, 9, np.array([9, 3, 6, 1]), seed=4)
    durs = np.array([0.6, 1e-5, 0.25, 1e-5], np.float32)  # int(1e-5 * 44100) = 0 samples -> 0 latent frames (cpp/helper.cpp:764-768)
    outs = {}
    for packed in (False, True):
        eng.set_packed_rows(packed)
        w, d = eng.synthesize(ids, mask, sttl, sdp, 3, 1.0, duration_override=durs, noise_seed=2)
        lat = eng.batch_fetch_latent()
        assert np.all(np.isfinite(w)) and np.all(lat[1] == 0) and np.all(lat[3] == 0)
        outs[packed] = (w, lat)
    tol = {"f32": 2e-5, "bf16": 2e-2, "f16": 2.5e-3}[dtype]
    assert rel_err(outs[True][1], outs[False][1])[0] < tol
    assert rel_err(outs[True][0], outs[False][0])[0] < 5 * tol
    np.testing.assert_array_equal(outs[True][0][1], outs[True][0][3])  # two silent utterances: identical output
    eng.set_packed_rows(True)


@pytest.mark.parametrize("dtype16", ["bf16", "f16"])
def test_trimmed_dense_vocoder_is_bit_identical(dtype16):
    """Default (reference) vocoder semantics: the padding is decoded as zero latent.  Where that padding is longer than twice the
    receptive field the engine computes only len*6 + 114 frames of the utterance and fills the position-independent rest from
    the model's cached zero-latent response — the returned [B, W] array must equal the dense computation BIT FOR BIT."""
    a = default_arch()
    eng = binding.Engine(0, dtype16)
    eng.load_synthetic(a, 7)
    # this test is about the VOCODER's two forms; the estimator's K4-split exists in the packed layout only (its fold kernels index
    # sequences through the packed row map), so it is switched off here to keep both layouts on the same estimator kernels
    eng.set_fused_ffn(1)
    rng = np.random.default_rng(3)
    for case in range(4):
        B = [6, 3, 12, 3][case]
        Lt = 48
        lens = rng.integers(2, Lt + 1, B)
        lens[0] = Lt
        ids, mask, sttl, sdp = make_inputs(a, B, Lt, lens, seed=40 + case)
        durs = rng.uniform(0.1, 1.0, B).astype(np.float32)
        durs[0] = [4.0, 3.0, 6.0, 4.5][case]  # one long utterance: everybody else gets a long zero-latent tail
        if case == 3:
            durs[1] = durs[0] - 0.5  # padding shorter than the receptive fields: this one stays dense, the third is trimmed
        outs = {}
        for packed in (False, True):
            eng.set_packed_rows(packed)
            for _ in range(3):  # eager, capture, replay
                w, d = eng.synthesize(ids, mask, sttl, sdp, 2, 1.0, duration_override=durs, noise_seed=9)
                if packed in outs:
                    np.testing.assert_array_equal(w, outs[packed][0])
            B_, L, W = eng.batch_dims()
            outs[packed] = (w, eng.vo_rows)
            assert (eng.vo_rows < B * L * 6) == packed, (case, packed, eng.vo_rows, B * L * 6)
        np.testing.assert_array_equal(outs[True][0], outs[False][0])
        assert outs[True][1] < outs[False][1]
    eng.set_packed_rows(True)


@pytest.mark.parametrize("dtype16", ["bf16", "f16"])
def test_trimmed_dense_vocoder_boundary_paddings(dtype16):
    """Paddings right at the trimming threshold: exactly 2 x 57 = 114 frames (no quiet region at all, computed frames meet the
    cached edge tail), 120 frames (6 quiet frames), and 108 frames (not trimmed): all bit-identical to the dense computation."""
    a = default_arch()
    eng = binding.Engine(0, dtype16)
    eng.load_synthetic(a, 7)
    eng.set_fused_ffn(1)  # (as above: the same estimator kernels in both layouts)
    # latent lengths 60 (longest), 41 (19 latent = 114 vocoder frames of padding), 40 (120), 42 (108: dense), 10
    durs = np.array([4.17, 2.85, 2.78, 2.92, 0.69], np.float32)
    B, Lt = 5, 30
    ids, mask, sttl, sdp = make_inputs(a, B, Lt, np.array([30, 20, 25, 12, 7]), seed=77)
    outs = {}
    for packed in (False, True):
        eng.set_packed_rows(packed)
        w, d = eng.synthesize(ids, mask, sttl, sdp, 2, 1.0, duration_override=durs, noise_seed=4)
        lens = host.latent_geometry(d, a.sample_rate, a.base_chunk_size, a.chunk_compress_factor, a.latent_dim)[2]
        assert list(lens) == [60, 41, 40, 42, 10], lens
        outs[packed] = (w, eng.vo_rows)
    np.testing.assert_array_equal(outs[True][0], outs[False][0])
    T = 60 * 6
    assert outs[False][1] == B * T
    assert outs[True][1] == T + (41 * 6 + 114) + (40 * 6 + 114) + T + (10 * 6 + 114), outs[True][1]
    eng.set_packed_rows(True)
