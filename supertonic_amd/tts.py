"""Python host with the names of the reference's `py/helper.py` (Style, TextToSpeech, load_text_to_speech,
load_voice_style) on top of the C ABI — what `py/service.py` and `py/example_onnx.py` import.

Differences from the reference are confined to where the work happens: the four `InferenceSession.run` sites
(/root/reference/py/helper.py:190-214) are one resident-batch call into the MI355X engine, the text frontend is the C++ one
(the C++ host is the contract where the reference's hosts disagree), and the chunks of a long text go
through the engine as ONE batch (length-aware vocoder) instead of one `_infer` per chunk (py/helper.py:231-243)."""
import json
import os
import secrets
import threading

import numpy as np

from . import binding, host
from .arch import default_arch


class Style:
    """py/helper.py:134-137"""

    def __init__(self, style_ttl: np.ndarray, style_dp: np.ndarray):
        self.ttl = np.ascontiguousarray(style_ttl, np.float32)
        self.dp = np.ascontiguousarray(style_dp, np.float32)


def load_voice_style(voice_style_paths, verbose=False, synthetic_arch=None):
    """py/helper.py:339-368.  With `synthetic_arch` (an engine running on synthetic weights, no assets on disk) a missing
    file yields a deterministic style keyed by the file's base name instead of an error."""
    ttl, dp = [], []
    for path in voice_style_paths:
        if not os.path.exists(path) and synthetic_arch is not None:
            a = synthetic_arch
            name = os.path.splitext(os.path.basename(path))[0]
            rng = np.random.default_rng(int.from_bytes(name.encode()[:8].ljust(8, b"\0"), "little"))
            ttl.append((rng.standard_normal((a.n_style_ttl, a.d_style_ttl)) * 0.1).astype(np.float32))
            dp.append((rng.standard_normal((a.n_style_dp, a.d_style_dp)) * 0.1).astype(np.float32))
            continue
        with open(path, "r") as f:
            vs = json.load(f)
        td, dd = vs["style_ttl"]["dims"], vs["style_dp"]["dims"]
        ttl.append(np.asarray(vs["style_ttl"]["data"], np.float32).reshape(td[1], td[2]))
        dp.append(np.asarray(vs["style_dp"]["data"], np.float32).reshape(dd[1], dd[2]))
    if len({t.shape for t in ttl}) != 1 or len({d.shape for d in dp}) != 1:
        raise ValueError("voice styles of one batch must share their dimensions")
    if verbose:
        print(f"Loaded {len(ttl)} voice styles")
    return Style(np.stack(ttl), np.stack(dp))


class TextToSpeech:
    """py/helper.py:140-258: `tts(text, lang, style, total_step, speed, silence_duration)` and `tts.batch(...)`, both
    returning (wav [B, W] float32, duration [B] float32).  One instance = one engine handle = one GPU; calls are
    serialised by a lock (the handle is single-threaded by contract)."""

    def __init__(self, engine, text_processor, cfgs, noise_seed=None):
        self.engine = engine
        self.text_processor = text_processor
        self.cfgs = cfgs
        self.sample_rate = cfgs["ae"]["sample_rate"]
        self.base_chunk_size = cfgs["ae"]["base_chunk_size"]
        self.chunk_compress_factor = cfgs["ttl"]["chunk_compress_factor"]
        self.ldim = cfgs["ttl"]["latent_dim"]
        self.noise_seed = noise_seed  # None: fresh noise per call, like np.random.randn in the reference
        self._lock = threading.Lock()
        self._calls = 0

    def _seed(self):
        if self.noise_seed is None:
            return secrets.randbits(63) | 1
        self._calls += 1
        return self.noise_seed + self._calls - 1

    def _infer(self, text_list, lang_list, style, total_step, speed=1.05, length_aware=False):
        if len(text_list) != style.ttl.shape[0]:
            raise ValueError("Number of texts must match number of style vectors")
        ids, mask = self.text_processor(text_list, lang_list)
        with self._lock:
            # length-aware batches (the chunks of a long text, the service's merged requests) come in ever-changing lengths: shape
            # buckets let them share captured graphs (stn_set_shape_buckets: every row stays exact over its own samples, rows are
            # just longer); a plain batch keeps the reference's exact [B, L * chunk] result
            self.engine.set_vocoder_mode(length_aware)
            self.engine.set_shape_buckets(length_aware)
            try:
                return self.engine.synthesize(ids, mask, style.ttl, style.dp, total_step, speed, noise_seed=self._seed())
            finally:
                self.engine.set_vocoder_mode(False)
                self.engine.set_shape_buckets(False)

    def latent_lengths(self, durations):
        """Latent frames each utterance occupies (get_latent_mask, py/helper.py:276-282) from its returned duration."""
        _, _, lens = host.latent_geometry(np.asarray(durations, np.float32), self.sample_rate, self.base_chunk_size,
                                          self.chunk_compress_factor, self.ldim)
        return np.asarray(lens)

    def solo_batch(self, text_list, lang_list, style, total_step, speed=1.05):
        """Independent utterances as one batch whose rows equal what each would give alone (length-aware vocoder):
        returns a list of per-utterance waves of L_i * chunk_size samples and the durations.  The building block of the
        long-form path and of the service's dynamic batching."""
        wav, dur = self._infer(text_list, lang_list, style, total_step, speed, length_aware=True)
        cs = self.base_chunk_size * self.chunk_compress_factor
        lens = [int(self.latent_lengths(dur[i:i + 1])[0]) for i in range(len(text_list))]
        return [wav[i, : min(n * cs, wav.shape[1])] for i, n in enumerate(lens)], dur

    def __call__(self, text, lang, style, total_step, speed=1.05, silence_duration=0.3):
        if style.ttl.shape[0] != 1:
            raise ValueError("Single speaker text to speech only supports single style")
        chunks = host.chunk_text(text, 120 if lang == "ko" else 300)
        if len(chunks) == 1:
            return self._infer(chunks, [lang], style, total_step, speed)
        n = len(chunks)
        rep = Style(np.repeat(style.ttl, n, axis=0), np.repeat(style.dp, n, axis=0))
        waves, dur = self.solo_batch(chunks, [lang] * n, rep, total_step, speed)
        silence = np.zeros(int(silence_duration * self.sample_rate), np.float32)
        parts, dur_cat = [], None
        for i, w in enumerate(waves):  # untrimmed chunk waves joined by zeros (py/helper.py:235-243)
            if i == 0:
                dur_cat = np.float32(dur[0])
            else:
                parts.append(silence)
                dur_cat = np.float32(dur_cat + np.float32(dur[i] + np.float32(silence_duration)))
            parts.append(w)
        return np.concatenate(parts)[None, :], np.array([dur_cat], np.float32)

    def batch(self, text_list, lang_list, style, total_step, speed=1.05):
        return self._infer(text_list, lang_list, style, total_step, speed)


def load_cfgs(onnx_dir):
    with open(os.path.join(onnx_dir, "tts.json"), "r") as f:
        return json.load(f)


def load_text_to_speech(onnx_dir, use_gpu=True, device=0, dtype="bf16", allow_synthetic=None, weight_seed=7, noise_seed=None):
    """py/helper.py:316-337.  use_gpu=True is the only mode (the reference only had the CPU one).  An unusable asset directory is an
    error, as in the reference (cpp/helper.cpp:805); only when the caller opts in — `allow_synthetic=True`, or TTS_ALLOW_SYNTHETIC=1
    in the environment when the argument is left at None — does the engine fall back to the default architecture on synthetic
    weights (benchmarks and tests on machines without the Hugging Face assets), and it says so."""
    if allow_synthetic is None:
        allow_synthetic = os.getenv("TTS_ALLOW_SYNTHETIC", "0").strip().lower() in {"1", "true", "yes", "y", "on"}
    if not use_gpu:
        raise NotImplementedError("CPU mode is not supported: this engine runs on MI355X only")
    eng = binding.Engine(device, dtype)
    try:
        eng.load_dir(onnx_dir)
        cfgs = load_cfgs(onnx_dir)
        with open(os.path.join(onnx_dir, "unicode_indexer.json"), "r") as f:
            tp = host.UnicodeProcessor(np.asarray(json.load(f), np.int64))
        synthetic = False
    except binding.StnError as e:
        if not allow_synthetic:
            raise
        print(f"model assets unavailable ({e}); synthetic weights from the default architecture (seed {weight_seed})")
        a = default_arch()
        eng.load_synthetic(a, weight_seed)
        cfgs = {"ae": {"sample_rate": a.sample_rate, "base_chunk_size": a.base_chunk_size},
                "ttl": {"chunk_compress_factor": a.chunk_compress_factor, "latent_dim": a.latent_dim}}
        tp = host.UnicodeProcessor(host.synthetic_indexer())
        synthetic = True
    tts = TextToSpeech(eng, tp, cfgs, noise_seed)
    tts.synthetic = synthetic
    return tts
