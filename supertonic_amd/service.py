"""HTTP face of the engine: the request schema, validation messages and responses of the reference's FastAPI service
(/root/reference/py/service.py:28-136: GET /health, POST /tts -> audio/wav or application/zip), on top of the C ABI.

What is different is throughput under concurrency.  The reference handles one request at a time on ORT-CPU.  Here
single-utterance requests that arrive together are merged by a `DynamicBatcher` into ONE resident batch on the GPU
(length-aware vocoder: a request's audio does not depend on who it shared the batch with) — the chunks of long texts
included; a `batch: true` request keeps the reference's semantics (its own padded batch, `TextToSpeech.batch`).

    TTS_ONNX_DIR=assets/onnx TTS_DTYPE=bf16 uvicorn supertonic_amd.service:app
"""
import io
import os
import threading
import time
import zipfile
from typing import List, Union

import numpy as np

from . import host
from .tts import Style, load_text_to_speech, load_voice_style

AVAILABLE_LANGS = host.AVAILABLE_LANGS


class _Job:
    __slots__ = ("texts", "lang", "style", "key", "done", "waves", "durs", "error")

    def __init__(self, texts, lang, style, key):
        self.texts, self.lang, self.style, self.key = texts, lang, style, key
        self.done = threading.Event()
        self.waves = self.durs = self.error = None


class DynamicBatcher:
    """Merges concurrent single-speaker jobs (each: the chunks of one text, one language, one style) that share
    (total_step, speed) into one engine batch.  A worker thread owns the engine: it takes the oldest job, waits up to
    `max_wait_ms` for company (or until `max_batch` utterances are queued), runs `tts.solo_batch` once and hands every job
    its own rows.  Rows are independent by construction, so merging changes latency and throughput, not audio."""

    def __init__(self, tts, max_batch=128, max_wait_ms=3.0):
        self.tts, self.max_batch, self.max_wait = tts, int(max_batch), max_wait_ms / 1e3
        self._q, self._cv = [], threading.Condition()
        self._stop = False
        self.batches = []  # sizes of the engine batches run so far (diagnostics / tests)
        self._t = threading.Thread(target=self._run, name="stn-batcher", daemon=True)
        self._t.start()

    def submit(self, texts, lang, style, total_step, speed):
        """Blocks until the job's utterances are synthesized; returns (list of waves, durations [n])."""
        job = _Job(list(texts), lang, style, (int(total_step), float(speed)))
        with self._cv:
            if self._stop:
                raise RuntimeError("batcher is closed")
            self._q.append(job)
            self._cv.notify_all()
        job.done.wait()
        if job.error is not None:
            raise job.error
        return job.waves, job.durs

    def close(self):
        with self._cv:
            self._stop = True
            self._cv.notify_all()
        self._t.join(timeout=10)

    def _take(self):
        with self._cv:
            while not self._q and not self._stop:
                self._cv.wait()
            if self._stop and not self._q:
                return None
            key = self._q[0].key
            deadline = time.monotonic() + self.max_wait
            while True:
                mine = [j for j in self._q if j.key == key]
                if sum(len(j.texts) for j in mine) >= self.max_batch or self._stop:
                    break
                left = deadline - time.monotonic()
                if left <= 0:
                    break
                self._cv.wait(left)
            picked, n = [], 0
            for j in [j for j in self._q if j.key == key]:
                if picked and n + len(j.texts) > self.max_batch:
                    break
                picked.append(j)
                n += len(j.texts)
            for j in picked:
                self._q.remove(j)
            return picked

    def _run(self):
        while True:
            jobs = self._take()
            if jobs is None:
                return
            try:
                texts = [t for j in jobs for t in j.texts]
                langs = [j.lang for j in jobs for _ in j.texts]
                ttl = np.concatenate([np.repeat(j.style.ttl, len(j.texts), axis=0) for j in jobs])
                dp = np.concatenate([np.repeat(j.style.dp, len(j.texts), axis=0) for j in jobs])
                step, speed = jobs[0].key
                waves, durs = self.tts.solo_batch(texts, langs, Style(ttl, dp), step, speed)
                self.batches.append(len(texts))
                o = 0
                for j in jobs:
                    n = len(j.texts)
                    j.waves, j.durs = waves[o:o + n], np.asarray(durs[o:o + n], np.float32)
                    o += n
            except Exception as e:  # the requests fail, the worker lives on
                for j in jobs:
                    j.error = e
            for j in jobs:
                j.done.set()


def join_chunks(waves, durs, silence_duration, sample_rate):
    """TextToSpeech.__call__'s concatenation (py/helper.py:235-243): untrimmed chunk waves with zeros between."""
    silence = np.zeros(int(silence_duration * sample_rate), np.float32)
    parts, dur = [], None
    for i, w in enumerate(waves):
        if i == 0:
            dur = np.float32(durs[0])
        else:
            parts.append(silence)
            dur = np.float32(dur + np.float32(durs[i] + np.float32(silence_duration)))
        parts.append(w)
    return np.concatenate(parts), float(dur)


def create_app(tts, max_batch=128, max_wait_ms=3.0, style_loader=None):
    """The FastAPI application around a TextToSpeech instance (supertonic_amd.tts or anything with its surface)."""
    from fastapi import FastAPI, HTTPException
    from fastapi.responses import JSONResponse, Response
    from pydantic import BaseModel, Field

    from contextlib import asynccontextmanager

    batcher = DynamicBatcher(tts, max_batch, max_wait_ms)

    @asynccontextmanager
    async def lifespan(_app):
        yield
        batcher.close()

    app = FastAPI(title="Supertonic TTS Service (MI355X)", lifespan=lifespan)
    app.state.batcher = batcher
    if style_loader is None:
        arch = tts.engine.arch if getattr(tts, "synthetic", False) else None

        def style_loader(paths):
            return load_voice_style(paths, verbose=False, synthetic_arch=arch)

    class TTSRequest(BaseModel):  # py/service.py:28-39
        text: Union[str, List[str]] = Field(..., description="Text to synthesize.")
        lang: Union[str, List[str]] = Field("en", description="Language(s) for text.")
        voice_style: Union[str, List[str]] = Field("assets/voice_styles/M1.json", description="Voice style path(s).")
        total_step: int = Field(5, ge=1, le=50)
        speed: float = Field(1.05, gt=0.0)
        batch: bool = False
        silence_duration: float = Field(0.3, ge=0.0, description="Silence between chunks for non-batch mode.")

    def ensure_list(v):
        return v if isinstance(v, list) else [v]

    @app.get("/health")
    def health():
        return JSONResponse({"status": "ok"})

    @app.post("/tts")
    def synthesize(req: TTSRequest):
        texts, langs, styles = ensure_list(req.text), ensure_list(req.lang), ensure_list(req.voice_style)
        if req.batch:
            if not (len(texts) == len(langs) == len(styles)):
                raise HTTPException(status_code=400, detail="text, lang, and voice_style must have the same length.")
        elif len(texts) != 1 or len(langs) != 1 or len(styles) != 1:
            raise HTTPException(status_code=400, detail="Non-batch mode requires single text, lang, and voice_style.")
        invalid = sorted({lang for lang in langs if lang not in AVAILABLE_LANGS})
        if invalid:
            raise HTTPException(status_code=400, detail=f"Invalid language(s): {', '.join(invalid)}")
        try:
            style = style_loader(styles)
        except (OSError, KeyError, ValueError) as e:
            raise HTTPException(status_code=400, detail=f"voice_style: {e}")
        sr = tts.sample_rate
        if req.batch:
            wav, dur = tts.batch(texts, langs, style, req.total_step, req.speed)
            chunks = [wav[i, : int(sr * float(dur[i]))] for i in range(wav.shape[0])]  # _slice_audio, py/service.py:62-71
        else:
            pieces = host.chunk_text(texts[0], 120 if langs[0] == "ko" else 300)
            waves, durs = batcher.submit(pieces, langs[0], style, req.total_step, req.speed)
            wav, d = join_chunks(waves, durs, req.silence_duration, sr)
            chunks = [wav[: int(sr * d)]]
        if len(chunks) == 1:
            name = host.sanitize_filename(texts[0], 40) or "tts"
            return Response(host.wav_bytes(chunks[0], sr), media_type="audio/wav",
                            headers={"Content-Disposition": f'attachment; filename="{_ascii(name)}.wav"'})
        zbuf = io.BytesIO()
        with zipfile.ZipFile(zbuf, "w", compression=zipfile.ZIP_DEFLATED) as zf:
            for i, c in enumerate(chunks):
                zf.writestr((host.sanitize_filename(texts[i], 40) or f"tts_{i + 1}") + ".wav", host.wav_bytes(c, sr))
        return Response(zbuf.getvalue(), media_type="application/zip",
                        headers={"Content-Disposition": 'attachment; filename="tts_outputs.zip"'})

    return app


def _ascii(name):
    """HTTP header values are latin-1: non-ASCII characters of a sanitized file name become '_' in the header only."""
    return "".join(ch if ord(ch) < 128 else "_" for ch in name)


def __getattr__(name):  # `uvicorn supertonic_amd.service:app` builds the engine on first use, not at import
    if name == "app":
        flag = os.getenv("TTS_USE_GPU", "1").strip().lower() in {"1", "true", "yes", "y", "on"}
        tts = load_text_to_speech(os.getenv("TTS_ONNX_DIR", "assets/onnx"), flag, int(os.getenv("TTS_DEVICE", "0")),
                                  os.getenv("TTS_DTYPE", "bf16"))
        globals()["app"] = create_app(tts, int(os.getenv("TTS_MAX_BATCH", "128")), float(os.getenv("TTS_MAX_WAIT_MS", "3")))
        return globals()["app"]
    raise AttributeError(name)
