"""Parity on the shapes of BASELINE.json's other configs, full 66 M stack, through the C ABI vs the CPU oracle:
  C4  mixed-length utterances (4..48 words) in one ragged batch, sharded as on 8 GPUs
  C5  multilingual batch (en/ko/es/pt/fr through the C++ text frontend: Hangul jamo + Latin decomposition) with an
      inference-steps sweep
Sizes are kept small enough for the oracle to finish in seconds; the full-size runs are property-checked instead
(finite, exact zero-masking, sharding invariance)."""
import numpy as np
import pytest

from oracle import host_ref
from oracle.neural_ref import RefModel, randn
from supertonic_amd import binding, host, workload
from supertonic_amd.arch import default_arch
from supertonic_amd.dist import shard_by_length
from gpu_util import parity_check, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref():
    return RefModel(default_arch(), 7)


@pytest.fixture(scope="module")
def eng_bf16():
    e = binding.Engine(0, "bf16")
    e.load_synthetic(default_arch(), 7)
    return e


@pytest.fixture(scope="module")
def eng_f16():
    e = binding.Engine(0, "f16")
    e.load_synthetic(default_arch(), 7)
    return e


@pytest.fixture(scope="module")
def eng_f32():
    e = binding.Engine(0, "f32")
    e.load_synthetic(default_arch(), 7)
    return e


def _prep(texts, langs, ids):
    a = default_arch()
    up = host.UnicodeProcessor(host.synthetic_indexer())
    tid, mask = up(texts, langs)
    sttl, sdp = workload.synthetic_styles(a, ids)
    return tid, mask, sttl, sdp


def test_c4_mixed_lengths_ragged_batch(ref, eng_f32, eng_bf16):
    texts = workload.utterances(6, min_words=4, max_words=48, seed=99)
    tid, mask, sttl, sdp = _prep(texts, ["en"] * 6, np.arange(6))
    durs = workload.forced_durations(texts)
    nz = {}

    def nf(B, D, L):
        nz["x"] = randn(1234, B, D, L, np.arange(6))
        return nz["x"]

    ref_wav, ref_dur = ref.synthesize(tid, mask, sttl, sdp, 3, 1.05, nf, duration_override=durs)
    for eng, mode in ((eng_f32, "f32"), (eng_bf16, "bf16")):
        wav, dur = eng.synthesize(tid, mask, sttl, sdp, 3, 1.05, noise=nz["x"], duration_override=durs)
        np.testing.assert_allclose(dur, ref_dur, rtol=1e-6)
        parity_check("c4.mixed_lengths_wav", mode, wav, ref_wav, "e2e")
        # latent beyond each utterance's own length is exactly zero (masked stages never leak into padding)
        lat = eng.batch_fetch_latent()
        _, L, lens = host.latent_geometry(dur, 44100, 512, 6, 24)
        for b in range(6):
            assert np.all(lat[b, :, lens[b]:] == 0) and np.abs(lat[b, :, :lens[b]]).min() >= 0


def test_c5_multilingual_steps_sweep(ref, eng_bf16, eng_f16):
    texts = ["Good morning to everyone here.", "안녕하세요 반갑습니다", "¿Cómo estás? Mañana será mejor", "Olá, você está bem? Ação",
             "Ça va très bien, merci à vous"]
    langs = ["en", "ko", "es", "pt", "fr"]
    tid, mask, sttl, sdp = _prep(texts, langs, np.arange(5))
    # the Korean utterance is tokenised as jamo, the accented ones as base + combining mark (cpp/helper.cpp:272-300)
    rids, rmask = host_ref.unicode_processor_call(host.synthetic_indexer().tolist(), texts, langs)
    assert np.array_equal(tid, rids) and np.array_equal(mask, rmask)
    durs = np.full(5, 1.2, np.float32)
    for steps in (2, 3, 5, 8, 10):  # the sweep SURVEY.md section 8 lists for config C5
        nz = {}

        def nf(B, D, L):
            nz["x"] = randn(7, B, D, L)
            return nz["x"]

        ref_wav, _ = ref.synthesize(tid, mask, sttl, sdp, steps, 1.0, nf, duration_override=durs)
        wav, _ = eng_bf16.synthesize(tid, mask, sttl, sdp, steps, 1.0, noise=nz["x"], duration_override=durs)
        mx, rms = parity_check(f"c5.multilingual_wav_steps{steps}", "bf16", wav, ref_wav, "e2e")  # no blow-up with more Euler steps
        # BASELINE config 5 as written: "fp16 MFMA linears" (STN_DTYPE_F16) — same stack, IEEE-half operands and activations
        wav16, _ = eng_f16.synthesize(tid, mask, sttl, sdp, steps, 1.0, noise=nz["x"], duration_override=durs)
        mx16, rms16 = parity_check(f"c5.multilingual_wav_steps{steps}", "f16", wav16, ref_wav, "e2e")
        assert rms16 < rms  # three more mantissa bits than bf16 must show


def test_c4_full_size_sharding_invariance(eng_bf16):
    """128 of the 1024 mixed-length utterances exactly as rank 3 of 8 would get them; a few are re-synthesized
    in a different batch composition and must give the same latent (masked stages, utterance-keyed noise)."""
    texts_all = workload.utterances(1024, min_words=4, max_words=48, seed=1234)
    shards = shard_by_length([len(t) for t in texts_all], 8)
    mine = shards[3]
    texts = [texts_all[i] for i in mine]
    tid, mask, sttl, sdp = _prep(texts, ["en"] * len(texts), mine)
    durs = workload.forced_durations(texts)
    wav, dur = eng_bf16.synthesize(tid, mask, sttl, sdp, 5, 1.05, duration_override=durs, noise_seed=1234, utt_ids=mine)
    assert wav.shape[0] == 128 and np.all(np.isfinite(wav))
    lat = eng_bf16.batch_fetch_latent()
    _, L, lens = host.latent_geometry(dur, 44100, 512, 6, 24)
    sub = [5, 64, 127]
    t2 = [texts[i] for i in sub]
    tid2, mask2, sttl2, sdp2 = _prep(t2, ["en"] * 3, mine[sub])
    eng_bf16.synthesize(tid2, mask2, sttl2, sdp2, 5, 1.05, duration_override=durs[sub], noise_seed=1234, utt_ids=mine[sub])
    lat2 = eng_bf16.batch_fetch_latent()
    for j, i in enumerate(sub):
        n = lens[i]
        mx, _ = rel_err(lat2[j, :, :n], lat[i, :, :n])
        assert mx < 2e-2, (i, mx)  # different tile shapes / summation orders in bf16, same math


def test_c4_full_size_all_eight_ranks_on_one_gpu(eng_bf16):
    """BASELINE.json configs[3] at FULL size through the native group path (include/stn_group.h): the 1024 mixed-length utterances
    (4..48 words) dealt over EIGHT ranks — length-sorted, round-robin, as on 8 GPUs — that share this box's one GPU (the rehearsal
    form: eight engines, eight streams and worker threads, the gather into rank 0 as device copies instead of RCCL sends), fetched in
    caller order.  Every utterance arrives with its duration, is finite and non-silent over its own samples; 24 utterances
    re-synthesized in ANOTHER batch composition (three from every rank, mixed into one batch on a single engine) give the same
    waveform to bf16 rounding — the independence the sharding relies on (/root/reference/cpp/helper.cpp:477: the batch is only a
    leading dimension)."""
    a = default_arch()
    texts_all = workload.utterances(1024, min_words=4, max_words=48, seed=1234)
    durs_all = workload.forced_durations(texts_all)
    tid, mask, sttl, sdp = _prep(texts_all, ["en"] * 1024, np.arange(1024))
    g = binding.Group([0] * 8, "bf16")
    g.load_synthetic(a, 7)
    pcm, dur = g.synthesize(tid, mask, sttl, sdp, 5, 1.05, duration_override=durs_all, noise_seed=1234)
    rows, samples = g.last_shards()
    g.close()
    assert list(rows) == [128] * 8 and pcm.shape == (1024, samples.max()) and pcm.dtype == np.int16
    np.testing.assert_allclose(dur, durs_all / np.float32(1.05), rtol=1e-6)
    rank_of, row_of = binding.group_deal(mask.sum(axis=(1, 2)).astype(np.int32), 8)
    ns = np.floor(dur * a.sample_rate).astype(int)
    for i in range(1024):
        own = pcm[i, :ns[i]]
        assert own.size > 1000 and np.abs(own.astype(np.int32)).max() > 0, i  # arrived and is not silence
        assert np.all(pcm[i, samples[rank_of[i]]:] == 0)                        # behind its shard's own row length: zeros
    # another composition: three utterances of every rank as ONE batch on a single engine (different lengths, neighbours, kernel regimes)
    ids2 = np.array(sorted(int(np.where((rank_of == r) & (row_of == j))[0][0]) for r in range(8) for j in (3, 64, 125)), dtype=np.int64)
    t2 = [texts_all[i] for i in ids2]
    tid2, mask2, sttl2, sdp2 = _prep(t2, ["en"] * len(t2), ids2)
    eng_bf16.batch_upload(tid2, mask2, sttl2, sdp2, duration_override=durs_all[ids2], utt_ids=ids2)
    eng_bf16.batch_run(5, 1.05, 1234)
    pcm2, dur2 = eng_bf16.batch_fetch_pcm16()
    np.testing.assert_allclose(dur2, dur[ids2], rtol=1e-6)
    worst = (0.0, 0.0)
    for j, i in enumerate(ids2):
        n = ns[i]
        x, y = pcm[i, :n].astype(np.float64), pcm2[j, :n].astype(np.float64)
        rms = np.sqrt(np.mean(x ** 2)) + 1.0
        mx, er = np.abs(x - y).max() / rms, np.sqrt(np.mean((x - y) ** 2)) / rms
        worst = (max(worst[0], mx), max(worst[1], er))
        # bf16, other tile shapes / kernel forms on both sides of the row thresholds, same math: the waveform of up to 20 s agrees to
        # a few percent rms (the latent to 4e-2 / 8e-3, tests/test_gpu_batch_invariance.py; the vocoder's 10 blocks carry it on)
        assert mx < 0.6 and er < 0.02, (int(i), mx, er)
    print(f"C4 another composition, PCM: worst max {worst[0]:.3f} rms {worst[1]:.4f} of the utterance's rms")


def test_c3_full_size_bench_workload_matches_oracle(ref, eng_f32, eng_bf16, eng_f16):
    """BASELINE.json configs[2] — exactly what bench.py times: 128 ten-word utterances, 5 Euler steps, the 66 M stack, forced
    durations, utterance-keyed Philox noise.  The fp32 CPU oracle does this batch in ~15 s on the box's cores, so the full
    size is checked directly rather than through properties: fp32 engine and bf16 engine against the oracle, then
    run-to-run determinism and eager == hipGraph replay on the bf16 engine."""
    a = default_arch()
    texts = workload.utterances(128, 10, seed=1234)
    ids_all = np.arange(128)
    tid, mask, sttl, sdp = _prep(texts, ["en"] * 128, ids_all)
    durs = workload.forced_durations(texts)
    nz = {}

    def nf(B, D, L):
        nz["x"] = randn(1234, B, D, L, ids_all.astype(np.int64))
        return nz["x"]

    rw, rd = ref.synthesize(tid, mask, sttl, sdp, 5, 1.05, nf, duration_override=durs)
    w32, d32 = eng_f32.synthesize(tid, mask, sttl, sdp, 5, 1.05, noise=nz["x"], duration_override=durs)
    np.testing.assert_allclose(d32, rd, rtol=1e-6)
    parity_check("c3.full_size_wav", "f32", w32, rw, "e2e")
    # bf16, device-side noise from the same (seed, utterance id) counters as the injected one
    w16, d16 = eng_bf16.synthesize(tid, mask, sttl, sdp, 5, 1.05, duration_override=durs, noise_seed=1234, utt_ids=ids_all)
    assert w16.shape == rw.shape == (128, 78 * 3072)
    parity_check("c3.full_size_wav", "bf16", w16, rw, "e2e")
    # IEEE-half mode on the same batch (its vocoder runs the f16 instantiation of K4 at this size)
    wh, _ = eng_f16.synthesize(tid, mask, sttl, sdp, 5, 1.05, duration_override=durs, noise_seed=1234, utt_ids=ids_all)
    parity_check("c3.full_size_wav", "f16", wh, rw, "e2e")
    # per-utterance: no single utterance is an outlier hidden by the batch rms
    worst = max(range(128), key=lambda i: rel_err(w16[i], rw[i])[1])
    parity_check("c3.full_size_worst_utterance_wav", "bf16", w16[worst], rw[worst], "e2e")
    # exact zeros past every utterance's own latent length are NOT produced by the reference's padded vocoder (zero latent
    # is signal), but the latent itself is masked exactly
    lat = eng_bf16.batch_fetch_latent()
    _, L, lens = host.latent_geometry(d16, 44100, 512, 6, 24)
    assert L == 78 and all(np.all(lat[i, :, lens[i]:] == 0) for i in range(128))
    # determinism and replay: the second run captures the graph, the third and fourth replay it
    r0 = eng_bf16.graph_replays
    for _ in range(3):
        w2, _ = eng_bf16.synthesize(tid, mask, sttl, sdp, 5, 1.05, duration_override=durs, noise_seed=1234, utt_ids=ids_all)
        assert np.array_equal(w2, w16)
    assert eng_bf16.graph_replays >= r0 + 2
