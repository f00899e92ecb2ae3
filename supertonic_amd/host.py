"""Python face of the C++ host text path (include/stn_host.h in libstn.so): same names as the reference's
Python twin (`/root/reference/py/helper.py`), same results as its C++ host (`cpp/helper.cpp`)."""
import ctypes

import numpy as np

from . import binding

AVAILABLE_LANGS = ["en", "ko", "es", "pt", "fr"]
_READY = False


def _lib():
    global _READY
    L = binding.load()
    if not _READY:
        c, sz, i64 = ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int64
        L.stn_host_last_error.restype = c
        L.stn_text_preprocess.restype = i64
        L.stn_text_preprocess.argtypes = [c, c, ctypes.c_void_p, sz]
        L.stn_text_to_ids.argtypes = [ctypes.c_void_p, sz, ctypes.POINTER(c), ctypes.POINTER(c), ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        L.stn_latent_geometry.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_int)] * 2 + [ctypes.c_void_p]
        L.stn_chunk_text.restype = i64
        L.stn_chunk_text.argtypes = [c, ctypes.c_int, ctypes.c_void_p, sz, ctypes.POINTER(ctypes.c_int)]
        L.stn_sanitize_filename.restype = i64
        L.stn_sanitize_filename.argtypes = [c, ctypes.c_int, ctypes.c_void_p, sz]
        L.stn_onnx_summary.restype = i64
        L.stn_onnx_summary.argtypes = [c, ctypes.c_void_p, sz]
        L.stn_bind_graphs.restype = i64
        L.stn_bind_graphs.argtypes = [c, ctypes.c_void_p, sz]
        L.stn_wav_encode.restype = i64
        L.stn_wav_encode.argtypes = [ctypes.c_void_p, sz, ctypes.c_int, ctypes.c_void_p, sz]
        L.stn_write_wav.argtypes = [c, ctypes.c_void_p, sz, ctypes.c_int]
        L.stn_load_voice_style.argtypes = [ctypes.POINTER(c), ctypes.c_int, ctypes.c_void_p, sz, ctypes.c_void_p, sz,
                                           ctypes.POINTER(ctypes.c_int64)]
        _READY = True
    return L


def _enc(s: str) -> bytes:
    return s.encode("utf-8", errors="surrogateescape")


def _dec(b: bytes) -> str:
    return b.decode("utf-8", errors="surrogateescape")


def _fail(L):
    raise ValueError(L.stn_host_last_error().decode())


def preprocess_text(text: str, lang: str) -> str:
    L = _lib()
    n = L.stn_text_preprocess(_enc(text), _enc(lang), None, 0)
    if n < 0:
        _fail(L)
    buf = ctypes.create_string_buffer(n + 1)
    L.stn_text_preprocess(_enc(text), _enc(lang), buf, n + 1)
    return _dec(buf.raw[:n])


class UnicodeProcessor:
    """texts -> (text_ids int64 [B,Lt], text_mask float32 [B,1,Lt]); mirrors UnicodeProcessor::call."""

    def __init__(self, indexer):
        self.indexer = np.ascontiguousarray(indexer, dtype=np.int64)

    def __call__(self, text_list, lang_list):
        L = _lib()
        B = len(text_list)
        if B == 0 or B != len(lang_list):
            raise ValueError("text_list and lang_list must be non-empty and of equal length")
        T = (ctypes.c_char_p * B)(*[_enc(t) for t in text_list])
        G = (ctypes.c_char_p * B)(*[_enc(g) for g in lang_list])
        lens = np.zeros(B, np.int32)
        lt = ctypes.c_int(0)
        if L.stn_text_to_ids(self.indexer.ctypes.data, self.indexer.size, T, G, B, None, 0, lens.ctypes.data, ctypes.byref(lt)) != 0:
            _fail(L)
        ids = np.zeros((B, lt.value), np.int64)
        if L.stn_text_to_ids(self.indexer.ctypes.data, self.indexer.size, T, G, B, ids.ctypes.data, lt.value, lens.ctypes.data, ctypes.byref(lt)) != 0:
            _fail(L)
        mask = (np.arange(lt.value)[None, :] < lens[:, None]).astype(np.float32).reshape(B, 1, lt.value)
        return ids, mask


def synthetic_indexer() -> np.ndarray:
    """Stand-in for unicode_indexer.json (absent): flat int64 table over UTF-16 code units, vocab 512."""
    cp = np.arange(65536)
    return np.where(cp < 384, cp, 384 + (cp % 128)).astype(np.int64)


def latent_geometry(duration, sample_rate, base_chunk_size, chunk_compress_factor, latent_dim):
    L = _lib()
    d = np.ascontiguousarray(duration, np.float32)
    lens = np.zeros(len(d), np.int32)
    D, Ln = ctypes.c_int(), ctypes.c_int()
    if L.stn_latent_geometry(d.ctypes.data, len(d), sample_rate, base_chunk_size, chunk_compress_factor, latent_dim,
                             ctypes.byref(D), ctypes.byref(Ln), lens.ctypes.data) != 0:
        _fail(L)
    return D.value, Ln.value, lens


def chunk_text(text: str, max_len: int = 300):
    L = _lib()
    n_chunks = ctypes.c_int()
    n = L.stn_chunk_text(_enc(text), max_len, None, 0, ctypes.byref(n_chunks))
    if n < 0:
        _fail(L)
    buf = ctypes.create_string_buffer(max(n, 1))
    L.stn_chunk_text(_enc(text), max_len, buf, n, ctypes.byref(n_chunks))
    parts = buf.raw[:n].split(b"\0")[:n_chunks.value]
    return [_dec(p) for p in parts]


def sanitize_filename(text: str, max_len: int) -> str:
    L = _lib()
    n = L.stn_sanitize_filename(_enc(text), max_len, None, 0)
    if n < 0:
        _fail(L)
    buf = ctypes.create_string_buffer(n + 1)
    L.stn_sanitize_filename(_enc(text), max_len, buf, n + 1)
    return _dec(buf.raw[:n])


def onnx_summary(path: str) -> dict:
    """What the built-in protobuf reader sees in an .onnx file (inputs, outputs, op histogram, initializers)."""
    import json
    L = _lib()
    n = L.stn_onnx_summary(path.encode(), None, 0)
    if n < 0:
        raise OSError(L.stn_host_last_error().decode())
    buf = ctypes.create_string_buffer(n + 1)
    L.stn_onnx_summary(path.encode(), buf, n + 1)
    return json.loads(buf.value.decode())


def bind_graphs(onnx_dir: str) -> dict:
    """The manifest-less binding of an asset directory (descriptor from the graphs' weight shapes, canonical tensor -> initializer),
    computed on the host only; raises OSError with the loader's diff when the graphs are not the engine's layout."""
    import json
    L = _lib()
    n = L.stn_bind_graphs(onnx_dir.encode(), None, 0)
    if n < 0:
        raise OSError(L.stn_host_last_error().decode())
    buf = ctypes.create_string_buffer(n + 1)
    L.stn_bind_graphs(onnx_dir.encode(), buf, n + 1)
    return json.loads(buf.value.decode())


def bound_tensor(onnx_dir: str, name: str) -> np.ndarray:
    """One canonical tensor of the manifest-less binding, flat fp32, as the engine would load it (host only)."""
    L = _lib()
    L.stn_bound_tensor.restype = ctypes.c_int64
    L.stn_bound_tensor.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t]
    n = L.stn_bound_tensor(onnx_dir.encode(), name.encode(), None, 0)
    if n < 0:
        raise OSError(L.stn_host_last_error().decode())
    out = np.empty(n, np.float32)
    L.stn_bound_tensor(onnx_dir.encode(), name.encode(), out.ctypes.data, n)
    return out


def wav_bytes(audio, sample_rate: int) -> bytes:
    L = _lib()
    a = np.ascontiguousarray(audio, np.float32)
    n = 44 + 2 * a.size
    buf = ctypes.create_string_buffer(n)
    if L.stn_wav_encode(a.ctypes.data, a.size, sample_rate, buf, n) != n:
        _fail(L)
    return buf.raw


def write_wav_file(path: str, audio, sample_rate: int):
    L = _lib()
    a = np.ascontiguousarray(audio, np.float32)
    if L.stn_write_wav(path.encode(), a.ctypes.data, a.size, sample_rate) != 0:
        raise OSError(L.stn_host_last_error().decode())


def load_voice_style_native(paths):
    """loadVoiceStyle of the C++ host (cpp/helper.cpp:829-897) through the C ABI -> (ttl [n, d1, d2], dp [n, e1, e2])."""
    L = _lib()
    arr = (ctypes.c_char_p * len(paths))(*[_enc(p) for p in paths])
    dims = (ctypes.c_int64 * 6)()
    if L.stn_load_voice_style(arr, len(paths), None, 0, None, 0, dims) != 0:
        raise OSError(L.stn_host_last_error().decode())
    ttl = np.empty((dims[0], dims[1], dims[2]), np.float32)
    dp = np.empty((dims[3], dims[4], dims[5]), np.float32)
    if L.stn_load_voice_style(arr, len(paths), ttl.ctypes.data, ttl.size, dp.ctypes.data, dp.size, dims) != 0:
        raise OSError(L.stn_host_last_error().decode())
    return ttl, dp
