"""ORACLE (test infrastructure): ctypes wrapper over oracle/libstnref.so (oracle/stn_ref.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes
import os
import subprocess

import numpy as np

from supertonic_amd.arch import StnArch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def default_threads():
    """Threads the oracle may use: the CPU affinity of this process, capped at 16 (a GPU box's CPU share);
    the host may have hundreds of cores the container is not allowed to use."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("STN_ORACLE_THREADS", "16"))))


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libstnref.so")
        if not os.path.exists(path):
            build()
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        L = ctypes.CDLL(path)
        L.stnref_set_threads.argtypes = [ctypes.c_int]
        L.stnref_set_threads(default_threads())
        vp, ci = ctypes.c_void_p, ctypes.c_int
        L.stnref_create.restype = vp
        L.stnref_create.argtypes = [ctypes.POINTER(StnArch), ctypes.c_uint64]
        L.stnref_destroy.argtypes = [vp]
        L.stnref_param_count.restype = ctypes.c_int64
        L.stnref_param_count.argtypes = [vp]
        L.stnref_num_tensors.argtypes = [vp]
        L.stnref_tensor_name.restype = ctypes.c_char_p
        L.stnref_tensor_name.argtypes = [vp, ci]
        L.stnref_tensor.restype = ctypes.c_int64
        L.stnref_tensor.argtypes = [vp, ctypes.c_char_p, vp, ctypes.c_int64]
        L.stnref_duration.argtypes = [vp, ci, ci, _i64p, _f32p, _f32p, _f32p]
        L.stnref_text_enc.argtypes = [vp, ci, ci, _i64p, _f32p, _f32p, _f32p]
        L.stnref_vector_est.argtypes = [vp, ci, ci, ci] + [_f32p] * 8
        L.stnref_vocoder.argtypes = [vp, ci, ci, _f32p, _f32p]
        L.stnref_linear.argtypes = [_f32p, ctypes.c_int64, ci, _f32p, vp, ci, _f32p]
        L.stnref_layernorm.argtypes = [_f32p, ctypes.c_int64, ci, _f32p, _f32p, ctypes.c_float, _f32p]
        L.stnref_dwconv.argtypes = [_f32p, ci, ci, ci, _f32p, _f32p, ci, ci, _f32p]
        L.stnref_attention_core.argtypes = [_f32p, _f32p, _f32p] + [ci] * 5 + [vp, _f32p]
        L.stnref_randn.argtypes = [ctypes.c_uint64, ci, ci, ci, vp, _f32p]
        _LIB = L
    return _LIB


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def set_gelu_tanh(on):
    """Process-wide GELU form of the oracle: False = erf (default), True = the tanh approximation (graphs that spell GELU with Tanh)."""
    lib().stnref_set_gelu_tanh(int(bool(on)))


class RefModel:
    """fp32 CPU restatement of the four stages on synthetic weights (arch, seed)."""

    def __init__(self, arch: StnArch, seed: int = 7):
        self.arch = arch
        self._h = lib().stnref_create(ctypes.byref(arch), seed)

    def close(self):
        if getattr(self, "_h", None):
            lib().stnref_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def param_count(self):
        return lib().stnref_param_count(self._h)

    def tensor_names(self):
        return [lib().stnref_tensor_name(self._h, i).decode() for i in range(lib().stnref_num_tensors(self._h))]

    def tensor(self, name):
        n = lib().stnref_tensor(self._h, name.encode(), None, 0)
        if n < 0:
            raise KeyError(name)
        out = np.empty(n, np.float32)
        lib().stnref_tensor(self._h, name.encode(), out.ctypes.data, n)
        return out

    def duration(self, text_ids, style_dp, text_mask):
        B, Lt = text_ids.shape
        out = np.empty(B, np.float32)
        lib().stnref_duration(self._h, B, Lt, _c(text_ids, np.int64), _c(style_dp, np.float32),
                              _c(text_mask, np.float32), out)
        return out

    def text_enc(self, text_ids, style_ttl, text_mask):
        B, Lt = text_ids.shape
        out = np.empty((B, self.arch.te_out_dim, Lt), np.float32)
        lib().stnref_text_enc(self._h, B, Lt, _c(text_ids, np.int64), _c(style_ttl, np.float32),
                              _c(text_mask, np.float32), out)
        return out

    def vector_est(self, noisy, text_emb, style_ttl, text_mask, latent_mask, total_step, current_step):
        B, D, L = noisy.shape
        Lt = text_emb.shape[2]
        out = np.empty((B, D, L), np.float32)
        lib().stnref_vector_est(self._h, B, L, Lt, _c(noisy, np.float32), _c(text_emb, np.float32),
                                _c(style_ttl, np.float32), _c(text_mask, np.float32), _c(latent_mask, np.float32),
                                _c(total_step, np.float32), _c(current_step, np.float32), out)
        return out

    def vocoder(self, latent):
        B, D, L = latent.shape
        out = np.empty((B, L * self.arch.chunk_size), np.float32)
        lib().stnref_vocoder(self._h, B, L, _c(latent, np.float32), out)
        return out

    def synthesize(self, text_ids, text_mask, style_ttl, style_dp, total_step, speed, noise_fn,
                   duration_override=None):
        """Stage order of TextToSpeech::_infer (/root/reference/cpp/helper.cpp:469-683).
        noise_fn(B, D, L) -> float32 [B,D,L].  Returns (wav [B, L*cs], duration [B])."""
        from . import host_ref
        a = self.arch
        dur = self.duration(text_ids, style_dp, text_mask)
        if duration_override is not None:
            dur = np.asarray(duration_override, np.float32).copy()
        dur = (dur / np.float32(speed)).astype(np.float32)
        emb = self.text_enc(text_ids, style_ttl, text_mask)
        D, L, lat = host_ref.latent_geometry(dur, a.sample_rate, a.base_chunk_size, a.chunk_compress_factor,
                                             a.latent_dim)
        lmask = host_ref.length_to_mask(lat, L)
        B = len(dur)
        xt = (noise_fn(B, D, L) * lmask).astype(np.float32)
        ts = np.full(B, total_step, np.float32)
        for s in range(total_step):
            xt = self.vector_est(xt, emb, style_ttl, text_mask, lmask, ts, np.full(B, s, np.float32))
        return self.vocoder(xt), dur


def threads():
    return lib().stnref_get_threads()


def randn(seed, B, D, L, utt_ids=None):
    out = np.empty((B, D, L), np.float32)
    ids = None if utt_ids is None else _c(utt_ids, np.int64)
    lib().stnref_randn(seed, B, D, L, None if ids is None else ids.ctypes.data, out)
    return out
