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
    for steps in (2, 5, 8):
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


def test_c4_full_size_all_eight_shards_on_one_gpu(eng_bf16):
    """BASELINE.json configs[3] at FULL size: the 1024 mixed-length utterances (4..48 words), length-sorted and dealt round-robin
    over 8 ranks exactly as bench.py --mixed --gpus 8 deals them; the eight 128-utterance shards run one after another on this GPU
    and their 16-bit PCM lands in the blocks a gather into rank 0 would fill (GatherPlan.local: the root's receive buffers).  Every
    utterance arrives, is finite and non-silent over its own samples; utterances re-synthesized in ANOTHER batch composition (three
    from every shard, mixed into one batch) give the same latent — the independence the sharding relies on
    (/root/reference/cpp/helper.cpp:477: the batch is only a leading dimension)."""
    import torch
    from supertonic_amd.dist import GatherPlan
    a = default_arch()
    texts_all = workload.utterances(1024, min_words=4, max_words=48, seed=1234)
    shards = shard_by_length([len(t) for t in texts_all], 8)
    durs_all = workload.forced_durations(texts_all)
    # shapes first (what GatherPlan's one-time all-gather exchanges): B and W = L * chunk samples per shard
    shapes = []
    for r in range(8):
        _, L, _ = host.latent_geometry(durs_all[shards[r]] / np.float32(1.05), a.sample_rate, a.base_chunk_size, a.chunk_compress_factor, a.latent_dim)
        shapes.append((len(shards[r]), L * a.chunk_size))
    plan = GatherPlan.local(shapes, torch.device("cuda", 0), torch.int16)
    keep = {}
    for r in range(8):
        mine = shards[r]
        texts = [texts_all[i] for i in mine]
        tid, mask, sttl, sdp = _prep(texts, ["en"] * len(texts), mine)
        eng_bf16.batch_upload(tid, mask, sttl, sdp, duration_override=durs_all[mine], utt_ids=mine)
        eng_bf16.batch_run(5, 1.05, 1234)
        B, L, W = eng_bf16.batch_dims()
        assert (B, W) == shapes[r]
        eng_bf16.batch_copy_pcm16_device(plan.wav_ptr(r), plan.stride)
        _, dur = eng_bf16.batch_fetch(want_wav=False)
        plan.set_durations(torch.from_numpy(dur).cuda(), r)
        lat = eng_bf16.batch_fetch_latent()
        assert np.all(np.isfinite(lat))
        for j in (3, 64, 125):
            keep[int(mine[j])] = lat[j].copy()
    torch.cuda.synchronize()
    wavs, durs_g = plan.result(0)
    assert len(wavs) == 8 and sum(w.shape[0] for w in wavs) == 1024
    seen = 0
    for r in range(8):
        pcm, d = wavs[r].cpu().numpy(), durs_g[r].cpu().numpy()
        assert pcm.shape == shapes[r] and pcm.dtype == np.int16
        np.testing.assert_allclose(d, durs_all[shards[r]] / np.float32(1.05), rtol=1e-6)
        ns = np.floor(d * a.sample_rate).astype(int)
        for j in range(pcm.shape[0]):
            own = pcm[j, :ns[j]]
            assert own.size > 1000 and np.abs(own.astype(np.int32)).max() > 0  # arrived and is not silence
            seen += 1
    assert seen == 1024
    # another composition: the 24 kept utterances as ONE batch (different lengths, different neighbours, different kernel regimes)
    ids2 = np.array(sorted(keep), dtype=np.int64)
    t2 = [texts_all[i] for i in ids2]
    tid2, mask2, sttl2, sdp2 = _prep(t2, ["en"] * len(t2), ids2)
    eng_bf16.synthesize(tid2, mask2, sttl2, sdp2, 5, 1.05, duration_override=durs_all[ids2], noise_seed=1234, utt_ids=ids2)
    lat2 = eng_bf16.batch_fetch_latent()
    _, _, lens2 = host.latent_geometry(durs_all[ids2] / np.float32(1.05), a.sample_rate, a.base_chunk_size, a.chunk_compress_factor, a.latent_dim)
    for j, i in enumerate(ids2):
        n = lens2[j]
        mx, rms = rel_err(lat2[j, :, :n], keep[int(i)][:, :n])
        assert mx < 4e-2 and rms < 8e-3, (int(i), mx, rms)  # bf16: other tile shapes / kernel forms, same math
        assert np.all(keep[int(i)][:, n:] == 0)


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
