"""The HTTP face (supertonic_amd/service.py) against the contract of the reference's service
(/root/reference/py/service.py:28-136): schema, validation messages, WAV / zip responses — and the part the reference does
not have, the dynamic batcher.  Runs on CPU with a stand-in synthesizer (the engine itself is covered by the GPU tests)."""
import io
import struct
import threading
import zipfile

import numpy as np
import pytest
from fastapi.testclient import TestClient

from supertonic_amd import host, service
from supertonic_amd.tts import Style

SR = 44100


class FakeTTS:
    """Each utterance -> a constant wave whose value encodes the text length, 0.01 s per character."""
    sample_rate = SR

    def __init__(self):
        self.solo_calls, self.batch_calls = [], []
        self.fail_next = False

    def _one(self, text):
        dur = np.float32(0.01 * max(len(text), 1))
        n = (int(SR * dur) + 3071) // 3072 * 3072  # untrimmed: whole latent frames, like the engine
        return np.full(n, min(len(text), 99) / 100.0, np.float32), dur

    def solo_batch(self, texts, langs, style, total_step, speed):
        if self.fail_next:
            self.fail_next = False
            raise RuntimeError("engine fell over")
        assert style.ttl.shape[0] == len(texts) == len(langs)
        self.solo_calls.append(list(texts))
        ws, ds = zip(*[self._one(t) for t in texts])
        return list(ws), np.array(ds, np.float32)

    def batch(self, texts, langs, style, total_step, speed=1.05):
        self.batch_calls.append(list(texts))
        ws, ds = zip(*[self._one(t) for t in texts])
        W = max(len(w) for w in ws)
        wav = np.zeros((len(ws), W), np.float32)
        for i, w in enumerate(ws):
            wav[i, : len(w)] = w
        return wav, np.array(ds, np.float32)


def _styles(paths):
    return Style(np.zeros((len(paths), 2, 4), np.float32), np.zeros((len(paths), 2, 3), np.float32))


@pytest.fixture()
def client():
    tts = FakeTTS()
    app = service.create_app(tts, max_batch=8, max_wait_ms=300.0, style_loader=_styles)
    with TestClient(app) as c:
        c.tts = tts
        c.batcher = app.state.batcher
        yield c


def _parse_wav(b):
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE" and b[12:16] == b"fmt "
    fmt, ch, sr, _, _, bits = struct.unpack("<HHIIHH", b[20:36])
    assert (fmt, ch, bits) == (1, 1, 16) and b[36:40] == b"data"
    n = struct.unpack("<I", b[40:44])[0]
    return sr, np.frombuffer(b[44:44 + n], "<i2")


def test_health(client):
    r = client.get("/health")
    assert r.status_code == 200 and r.json() == {"status": "ok"}


def test_validation_messages_match_the_reference(client):
    r = client.post("/tts", json={"text": ["a", "b"]})
    assert r.status_code == 400 and r.json()["detail"] == "Non-batch mode requires single text, lang, and voice_style."
    r = client.post("/tts", json={"text": ["a", "b"], "lang": ["en"], "voice_style": ["x", "y"], "batch": True})
    assert r.status_code == 400 and r.json()["detail"] == "text, lang, and voice_style must have the same length."
    r = client.post("/tts", json={"text": ["a", "b"], "lang": ["xx", "de"], "voice_style": ["x", "y"], "batch": True})
    assert r.status_code == 400 and r.json()["detail"] == "Invalid language(s): de, xx"
    for bad in ({"text": "a", "total_step": 0}, {"text": "a", "total_step": 51}, {"text": "a", "speed": 0.0},
                {"text": "a", "silence_duration": -1.0}, {"lang": "en"}):
        assert client.post("/tts", json=bad).status_code == 422  # pydantic field constraints (py/service.py:34-39)


def test_single_request_returns_trimmed_wav(client):
    text = "Hello there, world!"
    r = client.post("/tts", json={"text": text})
    assert r.status_code == 200 and r.headers["content-type"] == "audio/wav"
    assert r.headers["content-disposition"] == 'attachment; filename="%s.wav"' % host.sanitize_filename(text, 40)
    sr, pcm = _parse_wav(r.content)
    dur = np.float32(0.01 * len(text))
    assert sr == SR and len(pcm) == int(SR * float(dur))  # wav[:int(sr * dur)]  (py/service.py:62-71)
    assert np.all(pcm == int(len(text) / 100.0 * 32767.0))
    assert client.tts.solo_calls == [[text]] and client.tts.batch_calls == []


def test_batch_request_keeps_reference_semantics_and_zips(client):
    texts = ["first one", "the second text is longer"]
    r = client.post("/tts", json={"text": texts, "lang": ["en", "ko"], "voice_style": ["a.json", "b.json"], "batch": True})
    assert r.status_code == 200 and r.headers["content-type"] == "application/zip"
    assert r.headers["content-disposition"] == 'attachment; filename="tts_outputs.zip"'
    zf = zipfile.ZipFile(io.BytesIO(r.content))
    assert zf.namelist() == [host.sanitize_filename(t, 40) + ".wav" for t in texts]
    for t, name in zip(texts, zf.namelist()):
        sr, pcm = _parse_wav(zf.read(name))
        assert sr == SR and len(pcm) == int(SR * float(np.float32(0.01 * len(t))))
    assert client.tts.batch_calls == [texts] and client.tts.solo_calls == []  # its own padded batch, not merged with others


def test_long_text_chunks_are_one_engine_batch_joined_by_silence(client):
    sent = "This sentence is exactly long enough to matter for the chunker, is it not? "
    text = (sent * 12).strip()
    pieces = host.chunk_text(text, 300)
    assert len(pieces) >= 3
    r = client.post("/tts", json={"text": text, "silence_duration": 0.25})
    assert r.status_code == 200
    assert client.tts.solo_calls == [pieces]  # all chunks in ONE call
    sr, pcm = _parse_wav(r.content)
    waves = [client.tts._one(p) for p in pieces]
    total = sum(len(w) for w, _ in waves) + (len(pieces) - 1) * int(0.25 * SR)
    dur = np.float32(waves[0][1])
    for _, d in waves[1:]:
        dur = np.float32(dur + np.float32(d + np.float32(0.25)))
    assert len(pcm) == min(total, int(SR * float(dur)))
    n0 = len(waves[0][0])
    assert np.all(pcm[n0: n0 + int(0.25 * SR)] == 0) and pcm[n0 - 1] != 0 and pcm[n0 + int(0.25 * SR)] != 0


def test_concurrent_requests_are_merged_and_each_gets_its_own_audio(client):
    texts = ["a" * n for n in (5, 11, 17, 23, 29, 35)]
    out = {}

    def go(t):
        out[t] = client.post("/tts", json={"text": t})

    th = [threading.Thread(target=go, args=(t,)) for t in texts]
    [t.start() for t in th]
    [t.join() for t in th]
    assert all(out[t].status_code == 200 for t in texts)
    for t in texts:
        _, pcm = _parse_wav(out[t].content)
        assert len(pcm) == int(SR * float(np.float32(0.01 * len(t)))) and np.all(pcm == int(len(t) / 100.0 * 32767.0))
    sizes = client.batcher.batches
    assert sum(sizes) == 6 and len(sizes) <= 2, sizes  # merged (300 ms window), not six batches of one
    # different (total_step, speed) never share a batch
    client.tts.solo_calls.clear()
    a = threading.Thread(target=lambda: client.post("/tts", json={"text": "xx", "total_step": 5}))
    b = threading.Thread(target=lambda: client.post("/tts", json={"text": "yyy", "total_step": 8}))
    a.start(); b.start(); a.join(); b.join()
    assert sorted(map(tuple, client.tts.solo_calls)) == [("xx",), ("yyy",)]


def test_engine_failure_fails_the_request_not_the_worker(client):
    client.tts.fail_next = True
    with pytest.raises(RuntimeError):
        client.post("/tts", json={"text": "boom"})
    r = client.post("/tts", json={"text": "fine again"})
    assert r.status_code == 200


def test_max_batch_caps_a_merge(client):
    texts = ["b" * (3 + i) for i in range(12)]
    th = [threading.Thread(target=lambda t=t: client.post("/tts", json={"text": t})) for t in texts]
    [t.start() for t in th]
    [t.join() for t in th]
    assert sum(client.batcher.batches) == 12 and max(client.batcher.batches) <= 8
