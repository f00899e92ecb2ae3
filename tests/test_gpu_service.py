"""The Python host (supertonic_amd/tts.py: the names of the reference's py/helper.py) and the HTTP service on a real engine."""
import struct
import threading

import numpy as np
import pytest

from supertonic_amd import host, service, tts as tts_mod
from gpu_util import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tts():
    t = tts_mod.load_text_to_speech("no_assets_here", use_gpu=True, dtype="bf16", noise_seed=11, allow_synthetic=True)
    assert t.synthetic and t.sample_rate == 44100
    return t


def _style(tts, names):
    return tts_mod.load_voice_style([f"assets/voice_styles/{n}.json" for n in names], synthetic_arch=tts.engine.arch)


def test_call_and_batch_shapes_follow_the_reference(tts):
    st = _style(tts, ["M1"])
    wav, dur = tts("Hello there, this is a short sentence.", "en", st, 5, 1.05)
    assert wav.ndim == 2 and wav.shape[0] == 1 and dur.shape == (1,) and wav.shape[1] % 3072 == 0
    assert wav.shape[1] >= int(44100 * float(dur[0])) and np.all(np.isfinite(wav))
    w2, d2 = tts.batch(["One.", "Two words here."], ["en", "ko"], _style(tts, ["M1", "F2"]), 3, 1.2)
    assert w2.shape[0] == 2 and d2.shape == (2,)
    with pytest.raises(ValueError, match="Number of texts must match number of style vectors"):
        tts.batch(["a", "b"], ["en", "en"], st, 2)
    with pytest.raises(ValueError, match="Single speaker text to speech only supports single style"):
        tts("x", "en", _style(tts, ["M1", "F1"]), 2)
    with pytest.raises(ValueError, match="Invalid language"):
        tts("x", "de", st, 2)


def test_long_form_is_one_batch_and_equals_chunk_by_chunk(tts):
    st = _style(tts, ["F1"])
    sent = "This sentence is exactly long enough to matter for the chunker, is it not? "
    text = (sent * 10).strip()
    pieces = host.chunk_text(text, 300)
    assert len(pieces) >= 3
    tts.noise_seed, tts._calls = 100, 0
    wav, dur = tts(text, "en", st, 4, 1.05, 0.3)
    # chunk by chunk, the reference's way (py/helper.py:231-243), with the noise each chunk had in the batch
    cs, sil = 3072, int(0.3 * 44100)
    parts, dcat = [], None
    for i, p in enumerate(pieces):
        ids, mask = tts.text_processor([p], ["en"])
        w, d = tts.engine.synthesize(ids, mask, st.ttl, st.dp, 4, 1.05, noise_seed=100, utt_ids=np.array([i], np.int64))
        if i:
            parts.append(np.zeros(sil, np.float32))
            dcat = np.float32(dcat + np.float32(d[0] + np.float32(0.3)))
        else:
            dcat = np.float32(d[0])
        parts.append(w[0])
    ref = np.concatenate(parts)
    assert wav.shape == (1, len(ref)) and dur[0] == dcat
    mx, rms = rel_err(wav[0], ref)
    assert mx < 3e-2, (mx, rms)  # bf16: batch-of-n vs batch-of-one tile shapes


def _parse_wav(b):
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE"
    sr = struct.unpack("<I", b[24:28])[0]
    n = struct.unpack("<I", b[40:44])[0]
    return sr, np.frombuffer(b[44:44 + n], "<i2")


def test_service_end_to_end_and_dynamic_batching(tts):
    from fastapi.testclient import TestClient
    tts.noise_seed, tts._calls = None, 0
    app = service.create_app(tts, max_batch=64, max_wait_ms=200.0)
    texts = ["Short one.", "A somewhat longer request for the service.", "Third request, medium length.",
             "안녕하세요 반갑습니다", "Le cinquième texte est en français."]
    langs = ["en", "en", "en", "ko", "fr"]
    with TestClient(app) as c:
        assert c.get("/health").json() == {"status": "ok"}
        out = {}

        def go(i):
            out[i] = c.post("/tts", json={"text": texts[i], "lang": langs[i], "voice_style": "assets/voice_styles/M2.json"})

        th = [threading.Thread(target=go, args=(i,)) for i in range(len(texts))]
        [t.start() for t in th]
        [t.join() for t in th]
        assert all(out[i].status_code == 200 and out[i].headers["content-type"] == "audio/wav" for i in out)
        sizes = list(app.state.batcher.batches)
        assert sum(sizes) == len(texts) and len(sizes) <= 2, sizes
        for i in out:
            sr, pcm = _parse_wav(out[i].content)
            assert sr == 44100 and len(pcm) > 4000 and np.abs(pcm).max() > 0
        r = c.post("/tts", json={"text": texts[:2], "lang": langs[:2], "voice_style": ["a.json", "b.json"], "batch": True})
        assert r.status_code == 200 and r.headers["content-type"] == "application/zip"
        assert c.post("/tts", json={"text": "x", "lang": "de"}).json()["detail"] == "Invalid language(s): de"


def test_missing_assets_fail_at_startup_by_default(monkeypatch):
    """The service must not serve audio from random weights when the asset directory does not load (py/helper.py raises too)."""
    from supertonic_amd import binding
    monkeypatch.delenv("TTS_ALLOW_SYNTHETIC", raising=False)
    with pytest.raises(binding.StnError):
        tts_mod.load_text_to_speech("no_assets_here", use_gpu=True, dtype="bf16")


def test_a_mixed_stream_of_requests_shares_a_few_captured_graphs(tts):
    """Twenty single-speaker requests of 6..14 words, one after another (each its own length-aware batch of one through solo_batch, the
    service's building block): with shape buckets the engine captures a handful of graphs for all of them instead of one per request
    shape, and replays from then on.  (/root/reference/py/service.py:79-136 runs every request through its own _infer.)"""
    from supertonic_amd import workload
    st = _style(tts, ["F1"])
    texts = workload.utterances(20, min_words=6, max_words=14, seed=77)
    eng = tts.engine
    shapes = set()
    for rnd in range(3):
        c0, r0 = eng.graphs_cached, eng.graph_replays
        for t in texts:
            waves, dur = tts.solo_batch([t], ["en"], st, 3, 1.05)
            assert len(waves) == 1 and np.all(np.isfinite(waves[0])) and waves[0].size > 1000
            B, L, W = eng.batch_dims()
            shapes.add((L, eng.ve_rows))
        if rnd == 2:
            assert eng.graph_replays - r0 == len(texts) and eng.graphs_cached == c0  # steady state: every request replays
    n_tokens = len({len(t) for t in texts})
    print(f"20 requests, {n_tokens} distinct text lengths -> {len(shapes)} (L, rows) buckets, {eng.graphs_cached} graphs cached")
    assert eng.graphs_cached <= 4 and n_tokens >= 10, (eng.graphs_cached, shapes)
