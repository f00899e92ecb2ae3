"""ctypes binding of include/stn.h (libstn.so).  The library is the product path: if it is missing or
no HIP device is present, loading / Engine() raises — there is no CPU fallback."""
import ctypes
import os

import numpy as np

from .arch import StnArch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STN_LIB") or os.path.join(_HERE, "libstn.so")  # STN_LIB: a diagnostic build (timing variants)
_LIB = None

F32, BF16, F16 = 0, 1, 2
_DTYPES = {"f32": F32, "fp32": F32, "float32": F32, "bf16": BF16, "f16": F16, "fp16": F16, "float16": F16, "half": F16,
           F32: F32, BF16: BF16, F16: F16}
ACT_NONE, ACT_GELU, ACT_SILU = 0, 1, 2


class StnError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"stn error {code}: {msg}")
        self.code = code


class StnConfig(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int32), ("dtype", ctypes.c_int32)]


_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def runtime_info(L=None):
    """{'hip_built': ..., 'hip_runtime': ..., 'torch_preloaded': ...}: which HIP runtime libstn.so was compiled against and which
    one this process bound it to (PyTorch-ROCm wheels bundle their own; importing torch first makes that the one in use)."""
    import sys
    L = L or load()
    b, r = ctypes.c_int(), ctypes.c_int()
    L.stn_hip_versions(ctypes.byref(b), ctypes.byref(r))
    return {"hip_built": b.value, "hip_runtime": r.value, "torch_preloaded": "torch" in sys.modules, "lib": LIB_PATH}


def device_count():
    """Visible HIP devices, through libstn.so (no PyTorch needed)."""
    n = load().stn_device_count()
    if n < 0:
        raise StnError(n, "hipGetDeviceCount failed")
    return n


def device_sync(device=0):
    """hipDeviceSynchronize on `device`, through libstn.so."""
    rc = load().stn_device_sync(int(device))
    if rc < 0:
        raise StnError(rc, "hipDeviceSynchronize failed")


def _check_runtime(L):
    """libstn.so's code objects are built and tested with the system ROCm; when the process bound it to another HIP runtime major.minor
    (torch's bundled copy), say so once: not an error — the GPU suite runs in exactly that configuration and
    records it (tests/test_gpu_runtime_record.py) — but a mismatch the user should be able to see."""
    import warnings
    info = runtime_info(L)
    b, r = info["hip_built"], info["hip_runtime"]
    if r > 0 and (b // 100000) != (r // 100000):  # HIP_VERSION = major * 10^7 + minor * 10^5 + patch
        warnings.warn(f"libstn.so was built against HIP {b // 10000000}.{b // 100000 % 100} but runs on HIP runtime {r // 10000000}.{r // 100000 % 100}"
                      + (" (PyTorch's bundled runtime was loaded first; set STN_NO_TORCH_PRELOAD=1 in processes that do not need torch)" if info["torch_preloaded"] else ""),
                      RuntimeWarning, stacklevel=3)


def load():
    """Load libstn.so (built by `make` / __graft_entry__.build()).  Raises if it is not there."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} not found: build it with `make` (hipcc --offload-arch=gfx950); "
                                "the engine has no fallback path")
    # PyTorch-ROCm wheels carry their own copy of the HIP runtime.  If libstn.so brings the system runtime in first, a later
    # `import torch` in the same process finds no GPU ("No HIP GPUs are available"); the other order works (libstn.so then
    # resolves against the runtime that is already loaded).  The multi-GPU helpers (supertonic_amd.dist) and bench.py need
    # torch in the same process, so when torch is installed it is imported before the library (STN_NO_TORCH_PRELOAD=1 skips this).
    import sys
    if "torch" not in sys.modules and os.environ.get("STN_NO_TORCH_PRELOAD") != "1" and os.environ.get("STN_HOST_ONLY") != "1":
        import importlib.util
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    if os.environ.get("STN_HOST_ONLY") == "1":
        # the sanitizer build of the host-side parsers (make host-asan, STN_LIB=build_asan/libstn_host_asan.so): include/stn_host.h
        # only, no engine entry points to declare (supertonic_amd.host sets its own signatures)
        _LIB = L
        return L
    vp, ci, cu64, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_float
    L.stn_hip_versions.argtypes = [ctypes.POINTER(ci), ctypes.POINTER(ci)]
    L.stn_ffn_fused_forms.argtypes = [ci, ci, ci]
    L.stn_device_count.argtypes = []
    L.stn_device_sync.argtypes = [ci]
    _check_runtime(L)
    L.stn_version.restype = ctypes.c_char_p
    L.stn_create.argtypes = [ctypes.POINTER(StnConfig), ctypes.POINTER(vp)]
    L.stn_destroy.argtypes = [vp]
    L.stn_last_error.restype = ctypes.c_char_p
    L.stn_last_error.argtypes = [vp]
    L.stn_load_dir.argtypes = [vp, ctypes.c_char_p]
    L.stn_tensor_names.argtypes = [vp, ctypes.POINTER(StnArch), ctypes.c_char_p, ctypes.c_size_t]
    L.stn_load_synthetic.argtypes = [vp, ctypes.POINTER(StnArch), cu64]
    L.stn_get_arch.argtypes = [vp, ctypes.POINTER(StnArch)]
    L.stn_param_count.restype = ctypes.c_int64
    L.stn_param_count.argtypes = [vp]
    L.stn_duration.argtypes = [vp, ci, ci, _i64p, _f32p, _f32p, _f32p]
    L.stn_text_enc.argtypes = [vp, ci, ci, _i64p, _f32p, _f32p, _f32p]
    L.stn_vector_est.argtypes = [vp, ci, ci, ci] + [_f32p] * 8
    L.stn_vocoder.argtypes = [vp, ci, ci, _f32p, _f32p]
    L.stn_batch_upload.argtypes = [vp, ci, ci, _i64p, _f32p, _f32p, _f32p, vp, vp]
    L.stn_batch_set_noise.argtypes = [vp, _f32p, ci]
    L.stn_batch_run.argtypes = [vp, ci, cf, cu64]
    L.stn_set_graph_mode.argtypes = [vp, ci]
    L.stn_set_vocoder_mode.argtypes = [vp, ci]
    L.stn_set_row_layout.argtypes = [vp, ci]
    L.stn_set_fused_xattn.argtypes = [vp, ci]
    L.stn_set_shape_buckets.argtypes = [vp, ci]
    L.stn_set_duration_read.argtypes = [vp, ci]
    L.stn_set_gelu_form.argtypes = [vp, ci]
    L.stn_get_gelu_form.argtypes = [vp]
    L.stn_batch_ve_rows.argtypes = [vp]
    L.stn_batch_ve_rows.restype = ctypes.c_int64
    L.stn_batch_vo_rows.argtypes = [vp]
    L.stn_batch_vo_rows.restype = ctypes.c_int64
    L.stn_graph_replays.restype = ctypes.c_int64
    L.stn_graph_replays.argtypes = [vp]
    L.stn_graphs_cached.restype = ctypes.c_int64
    L.stn_graphs_cached.argtypes = [vp]
    L.stn_batch_dims.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ctypes.c_int64)]
    L.stn_batch_fetch.argtypes = [vp, vp, ctypes.c_size_t, vp]
    L.stn_batch_fetch_latent.argtypes = [vp, _f32p]
    L.stn_batch_fetch_pcm16.argtypes = [vp, vp, ctypes.c_size_t, vp]
    L.stn_batch_fetch_pcm16_begin.argtypes = [vp, ci]
    L.stn_batch_fetch_slot_dims.argtypes = [vp, ci, ctypes.POINTER(ci), ctypes.POINTER(ctypes.c_int64)]
    L.stn_batch_fetch_pcm16_end.argtypes = [vp, ci, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t), vp]
    L.stn_host_alloc_pinned.restype = vp
    L.stn_host_alloc_pinned.argtypes = [ctypes.c_size_t]
    L.stn_host_free_pinned.argtypes = [vp]
    L.stn_batch_wav_device_ptr.argtypes = [vp, ctypes.POINTER(vp)]
    L.stn_sync.argtypes = [vp]
    L.stn_set_stream.argtypes = [vp, vp]
    L.stn_batch_copy_wav_device.argtypes = [vp, vp, ctypes.c_int64]
    L.stn_batch_copy_pcm16_device.argtypes = [vp, vp, ctypes.c_int64]
    L.stn_profile_filter.argtypes = [vp, ctypes.c_char_p]
    L.stn_profile_sample.argtypes = [vp, ci]
    L.stn_launch_log_enable.argtypes = [vp, ci]
    L.stn_dbg_xattn_hs_enable.argtypes = [vp, ci]
    L.stn_dbg_xattn_hs_stamps.argtypes = [vp, vp, ctypes.c_size_t]
    L.stn_dbg_xattn_hs_stamps.restype = ctypes.c_int64
    L.stn_dbg_fold_run_frames.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    L.stn_dbg_fold_run_frames.restype = ctypes.c_int
    L.stn_launch_log.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t]
    L.stn_launch_log.restype = ctypes.c_int64
    L.stn_profile_enable.argtypes = [vp, ci]
    L.stn_profile_reset.argtypes = [vp]
    L.stn_profile_count.argtypes = [vp]
    L.stn_profile_get.argtypes = [vp, ci, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_double),
                                  ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double),
                                  ctypes.POINTER(ctypes.c_double)]
    L.stn_op_gemm.argtypes = [vp, ci, ci, ci, ci, _f32p, _f32p, vp, ci, _f32p]
    L.stn_op_gemm_bench.argtypes = [vp, ci, ci, ci, ci, ci, ci, ctypes.POINTER(ctypes.c_double)]
    L.stn_op_gemm_phases.argtypes = [vp, ci, ci, ci, ci, ci, ctypes.POINTER(ctypes.c_double)]
    L.stn_op_dwconv_ln.argtypes = [vp, ci, ci, ci, ci, ci, ci, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p]
    L.stn_op_dwconv_ln_ragged.argtypes = [vp, ci, ci, ci, ci, ci, ci, _f32p, _f32p, _f32p, _f32p, _f32p, _i32p, _f32p]
    L.stn_op_attention.argtypes = [vp, ci, ci, ci, ci, ci, ci, _f32p, _f32p, _f32p, vp, vp, ci, _f32p]
    L.stn_op_randn.argtypes = [vp, cu64, ci, ci, ci, vp, vp, _f32p]
    L.stn_op_ffn.argtypes = [vp, ci, ci, ci, _f32p, _f32p, _f32p, _f32p, vp, vp, vp, vp, ci, _f32p, ci]
    L.stn_op_ffn_bench.argtypes = [vp, ci, ci, ci, ci, ci, ctypes.POINTER(ctypes.c_double)]
    L.stn_set_fused_ffn.argtypes = [vp, ci]
    L.stn_set_fused_ffn_min_rows.argtypes = [vp, ctypes.c_int64, ctypes.c_int64]
    L.stn_op_fold_dwconv_ln.argtypes = [vp, ci, ci, ci, ci, ci, _i32p, _f32p, _f32p, vp, vp, vp, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p]
    L.stn_op_block_bench.argtypes = [vp, ci, ci, ci, ci, ci, ci, ci, ci, ctypes.POINTER(ctypes.c_double)]
    # include/stn_group.h: several devices in one process
    L.stn_group_create.argtypes = [ci, vp, ci, ctypes.POINTER(vp)]
    L.stn_group_destroy.argtypes = [vp]
    L.stn_group_last_error.restype = ctypes.c_char_p
    L.stn_group_last_error.argtypes = [vp]
    L.stn_group_size.argtypes = [vp]
    L.stn_group_uses_rccl.argtypes = [vp]
    L.stn_group_handle.restype = vp
    L.stn_group_handle.argtypes = [vp, ci]
    L.stn_group_load_synthetic.argtypes = [vp, ctypes.POINTER(StnArch), cu64]
    L.stn_group_load_dir.argtypes = [vp, ctypes.c_char_p]
    L.stn_group_deal.argtypes = [ci, _i32p, ci, _i32p, _i32p]
    L.stn_group_synthesize.argtypes = [vp, ci, ci, _i64p, _f32p, _f32p, _f32p, ci, cf, vp, cu64, ctypes.POINTER(ctypes.c_int64)]
    L.stn_group_fetch_pcm16.argtypes = [vp, vp, ctypes.c_size_t, vp]
    L.stn_group_last_shards.argtypes = [vp, _i32p, np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")]
    _LIB = L
    return L


def group_deal(lengths, n_ranks):
    """include/stn_group.h's deal (host only): (rank_of, row_of) for utterances of these token counts."""
    lengths = _c(lengths, np.int32)
    rank_of, row_of = np.empty(len(lengths), np.int32), np.empty(len(lengths), np.int32)
    rc = load().stn_group_deal(len(lengths), lengths, int(n_ranks), rank_of, row_of)
    if rc < 0:
        raise StnError(rc, "stn_group_deal: invalid arguments")
    return rank_of, row_of


class Group:
    """n devices in one process (include/stn_group.h): one engine per device, utterances dealt by length, 16-bit PCM gathered into the
    first device (RCCL when the devices are distinct; the same ordinal repeated is a rehearsal on one GPU)."""

    def __init__(self, devices, dtype="bf16"):
        self._lib = load()
        self._g = ctypes.c_void_p()
        devices = list(range(devices)) if isinstance(devices, int) else list(devices)
        arr = (ctypes.c_int * len(devices))(*devices)
        rc = self._lib.stn_group_create(len(devices), arr, _DTYPES[dtype], ctypes.byref(self._g))
        if rc < 0:
            self._g = None
            raise StnError(rc, self._lib.stn_group_last_error(None).decode())
        self.n = len(devices)

    def close(self):
        if getattr(self, "_g", None):
            self._lib.stn_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc < 0:
            raise StnError(rc, self._lib.stn_group_last_error(self._g).decode())

    @property
    def uses_rccl(self):
        return bool(self._lib.stn_group_uses_rccl(self._g))

    def load_synthetic(self, arch: StnArch, seed: int = 7):
        self._ck(self._lib.stn_group_load_synthetic(self._g, ctypes.byref(arch), seed))

    def load_dir(self, onnx_dir: str):
        self._ck(self._lib.stn_group_load_dir(self._g, onnx_dir.encode()))

    def synthesize(self, text_ids, text_mask, style_ttl, style_dp, total_step=5, speed=1.05, duration_override=None, noise_seed=1234):
        """-> (pcm [B, W] int16 in caller order, duration [B])"""
        B, Lt = text_ids.shape
        _d, dptr = _opt(duration_override, np.float32)
        W = ctypes.c_int64()
        self._ck(self._lib.stn_group_synthesize(self._g, B, Lt, _c(text_ids, np.int64), _c(text_mask, np.float32), _c(style_ttl, np.float32),
                                                _c(style_dp, np.float32), total_step, speed, dptr, noise_seed, ctypes.byref(W)))
        pcm, dur = np.empty((B, W.value), np.int16), np.empty(B, np.float32)
        self._ck(self._lib.stn_group_fetch_pcm16(self._g, pcm.ctypes.data, pcm.size, dur.ctypes.data))
        return pcm, dur

    def last_shards(self):
        rows, samples = np.zeros(self.n, np.int32), np.zeros(self.n, np.int64)
        self._ck(self._lib.stn_group_last_shards(self._g, rows, samples))
        return rows, samples


def fold_run_frames(latent_lengths, n_cu=256):
    """frames per workgroup of the estimator's fold kernel for these latent lengths (0 = the default 32); host-only"""
    a = np.ascontiguousarray(latent_lengths, dtype=np.int32)
    r = load().stn_dbg_fold_run_frames(a.ctypes.data, len(a), int(n_cu))
    if r < 0:
        raise StnError(r, "stn_dbg_fold_run_frames: bad arguments")
    return int(r)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _opt(a, dt):
    if a is None:
        return None, None
    arr = _c(a, dt)
    return arr, arr.ctypes.data


class Engine:
    """One GPU + one HIP stream.  Mirrors the four ONNX sessions of the reference's TextToSpeech
    (/root/reference/cpp/helper.cpp:404-422) behind the C ABI."""

    def __init__(self, device=0, dtype="bf16"):
        self._lib = load()
        self._h = ctypes.c_void_p()
        cfg = StnConfig(device, _DTYPES[dtype])
        rc = self._lib.stn_create(ctypes.byref(cfg), ctypes.byref(self._h))
        if rc != 0:
            msg = self._lib.stn_last_error(None).decode()
            self._h = None
            raise StnError(rc, msg)
        self.dtype = _DTYPES[dtype]
        self.device = device
        self.arch = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.stn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _need_model(self):
        if self.arch is None:
            raise StnError(-3, "no model loaded: call load_synthetic or load_dir first")

    def _ck(self, rc):
        if rc < 0:
            raise StnError(rc, self._lib.stn_last_error(self._h).decode())
        return rc

    # ---- model -------------------------------------------------------------------------------
    def load_synthetic(self, arch: StnArch, seed: int = 7):
        self._ck(self._lib.stn_load_synthetic(self._h, ctypes.byref(arch), seed))
        self.arch = arch

    def last_error(self) -> str:
        """stn_last_error: the last failure's text — or, after a successful manifest-less stn_load_dir, its notes."""
        return (self._lib.stn_last_error(self._h) or b"").decode()

    def load_dir(self, onnx_dir: str):
        self._ck(self._lib.stn_load_dir(self._h, onnx_dir.encode()))
        a = StnArch()
        self._ck(self._lib.stn_get_arch(self._h, ctypes.byref(a)))
        self.arch = a

    def tensor_names(self, arch: StnArch):
        n = self._ck(self._lib.stn_tensor_names(self._h, ctypes.byref(arch), None, 0))
        buf = ctypes.create_string_buffer(n + 1)
        self._ck(self._lib.stn_tensor_names(self._h, ctypes.byref(arch), buf, n + 1))
        return buf.value.decode().split("\n")[:-1]

    @property
    def param_count(self):
        return self._lib.stn_param_count(self._h)

    # ---- the four former Run sites ----------------------------------------------------------------
    def duration(self, text_ids, style_dp, text_mask):
        self._need_model()
        B, Lt = text_ids.shape
        out = np.empty(B, np.float32)
        self._ck(self._lib.stn_duration(self._h, B, Lt, _c(text_ids, np.int64), _c(style_dp, np.float32),
                                        _c(text_mask, np.float32), out))
        return out

    def text_enc(self, text_ids, style_ttl, text_mask):
        self._need_model()
        B, Lt = text_ids.shape
        out = np.empty((B, self.arch.te_out_dim, Lt), np.float32)
        self._ck(self._lib.stn_text_enc(self._h, B, Lt, _c(text_ids, np.int64), _c(style_ttl, np.float32),
                                        _c(text_mask, np.float32), out))
        return out

    def vector_est(self, noisy, text_emb, style_ttl, text_mask, latent_mask, total_step, current_step):
        self._need_model()
        B, D, L = noisy.shape
        Lt = text_emb.shape[2]
        out = np.empty((B, D, L), np.float32)
        self._ck(self._lib.stn_vector_est(self._h, B, L, Lt, _c(noisy, np.float32), _c(text_emb, np.float32),
                                          _c(style_ttl, np.float32), _c(text_mask, np.float32),
                                          _c(latent_mask, np.float32), _c(total_step, np.float32),
                                          _c(current_step, np.float32), out))
        return out

    def vocoder(self, latent):
        self._need_model()
        B, D, L = latent.shape
        out = np.empty((B, L * self.arch.chunk_size), np.float32)
        self._ck(self._lib.stn_vocoder(self._h, B, L, _c(latent, np.float32), out))
        return out

    # ---- fused, HBM-resident synthesis ----------------------------------------------------------------
    def batch_upload(self, text_ids, text_mask, style_ttl, style_dp, duration_override=None, utt_ids=None):
        B, Lt = text_ids.shape
        _d, dptr = _opt(duration_override, np.float32)
        _u, uptr = _opt(utt_ids, np.int64)
        self._ck(self._lib.stn_batch_upload(self._h, B, Lt, _c(text_ids, np.int64), _c(text_mask, np.float32),
                                            _c(style_ttl, np.float32), _c(style_dp, np.float32), dptr, uptr))

    def batch_set_noise(self, noise):
        noise = _c(noise, np.float32)
        self._ck(self._lib.stn_batch_set_noise(self._h, noise, noise.shape[2]))

    def batch_run(self, total_step=5, speed=1.05, noise_seed=1234):
        self._ck(self._lib.stn_batch_run(self._h, total_step, speed, noise_seed))

    def set_graph_mode(self, on=True):
        self._ck(self._lib.stn_set_graph_mode(self._h, int(on)))

    def set_packed_rows(self, on=True):
        """Vector-estimator row layout in batch_run: packed (default, no work on padding) or padded [b*L + t]."""
        self._ck(self._lib.stn_set_row_layout(self._h, int(bool(on))))

    def set_fused_ffn(self, mask):
        """K4 stage mask: 1 vocoder, 2 vector estimator, 4 text encoder / duration predictor (0 = two GEMM launches everywhere)."""
        self._ck(self._lib.stn_set_fused_ffn(self._h, int(mask)))

    def set_shape_buckets(self, on=True):
        """Round Lt, L and the packed row counts up to bucket boundaries so that requests of unlike lengths share captured graphs."""
        self._ck(self._lib.stn_set_shape_buckets(self._h, int(bool(on))))

    def set_duration_read(self, always):
        """Measurement aid: read the predicted durations back even when they are overridden (the critical path of a predicted-duration run)."""
        self._ck(self._lib.stn_set_duration_read(self._h, int(bool(always))))

    def set_gelu_form(self, tanh_form):
        """0 erf (default), 1 the tanh approximation (what stn_load_dir selects for graphs that spell GELU with Tanh)."""
        self._ck(self._lib.stn_set_gelu_form(self._h, int(bool(tanh_form))))

    @property
    def gelu_form(self):
        return self._lib.stn_get_gelu_form(self._h)

    def set_fused_xattn(self, on=True):
        """Cross-attention blocks of the vector estimator: True head-split (default), False four launches."""
        self._ck(self._lib.stn_set_fused_xattn(self._h, int(on)))

    @property
    def vo_rows(self):
        return self._lib.stn_batch_vo_rows(self._h)

    @property
    def ve_rows(self):
        return self._lib.stn_batch_ve_rows(self._h)

    def set_vocoder_mode(self, length_aware):
        """False: the reference's batched vocoder (padding decoded as zero latent). True: every utterance ends at its own length."""
        self._ck(self._lib.stn_set_vocoder_mode(self._h, int(bool(length_aware))))

    def fetch_pcm16_begin(self, slot):
        """Start the PCM conversion + device->host copy of the finished batch on `slot` (0/1); returns at once."""
        self._ck(self._lib.stn_batch_fetch_pcm16_begin(self._h, int(slot)))

    def fetch_pcm16_end(self, slot, copy=True):
        """Wait for the slot's copy -> (pcm [B, W] int16, duration [B]).  copy=False returns a view of the handle's pinned buffer
        (valid until the slot's next fetch_pcm16_begin)."""
        B, W = ctypes.c_int(), ctypes.c_int64()
        self._ck(self._lib.stn_batch_fetch_slot_dims(self._h, int(slot), ctypes.byref(B), ctypes.byref(W)))  # the slot's batch, not the resident one
        ptr, n = ctypes.c_void_p(), ctypes.c_size_t()
        dur = np.empty(B.value, np.float32)
        self._ck(self._lib.stn_batch_fetch_pcm16_end(self._h, int(slot), ctypes.byref(ptr), ctypes.byref(n), dur.ctypes.data))
        arr = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_int16)), shape=(n.value,))
        rows = n.value // max(dur.size, 1)
        arr = arr.reshape(dur.size, rows)
        return (arr.copy() if copy else arr), dur

    @property
    def graph_replays(self):
        return self._lib.stn_graph_replays(self._h)

    @property
    def graphs_cached(self):
        return self._lib.stn_graphs_cached(self._h)

    def batch_dims(self):
        B, L, W = ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()
        self._ck(self._lib.stn_batch_dims(self._h, ctypes.byref(B), ctypes.byref(L), ctypes.byref(W)))
        return B.value, L.value, W.value

    def batch_fetch(self, want_wav=True):
        B, L, W = self.batch_dims()
        dur = np.empty(B, np.float32)
        wav = np.empty((B, W), np.float32) if want_wav else None
        self._ck(self._lib.stn_batch_fetch(self._h, wav.ctypes.data if want_wav else None, B * W if want_wav else 0,
                                           dur.ctypes.data))
        return wav, dur

    def batch_fetch_pcm16(self):
        B, L, W = self.batch_dims()
        pcm = np.empty((B, W), np.int16)
        dur = np.empty(B, np.float32)
        self._ck(self._lib.stn_batch_fetch_pcm16(self._h, pcm.ctypes.data, B * W, dur.ctypes.data))
        return pcm, dur

    def batch_fetch_latent(self):
        B, L, _ = self.batch_dims()
        out = np.empty((B, self.arch.latent_channels, L), np.float32)
        self._ck(self._lib.stn_batch_fetch_latent(self._h, out))
        return out

    def batch_wav_device_ptr(self):
        p = ctypes.c_void_p()
        self._ck(self._lib.stn_batch_wav_device_ptr(self._h, ctypes.byref(p)))
        return p.value

    def sync(self):
        self._ck(self._lib.stn_sync(self._h))

    def set_stream(self, hip_stream_ptr):
        """Enqueue on a caller-owned HIP stream (int pointer, e.g. torch.cuda.current_stream().cuda_stream)."""
        self._ck(self._lib.stn_set_stream(self._h, hip_stream_ptr))

    def batch_copy_pcm16_device(self, dst_ptr, dst_stride):
        self._ck(self._lib.stn_batch_copy_pcm16_device(self._h, dst_ptr, dst_stride))

    def batch_copy_wav_device(self, dst_ptr, dst_stride):
        self._ck(self._lib.stn_batch_copy_wav_device(self._h, dst_ptr, dst_stride))

    def synthesize(self, text_ids, text_mask, style_ttl, style_dp, total_step=5, speed=1.05, noise=None,
                   duration_override=None, noise_seed=1234, utt_ids=None):
        """TextToSpeech::_infer (/root/reference/cpp/helper.cpp:469-683) -> (wav [B, L*cs], duration [B])."""
        self.batch_upload(text_ids, text_mask, style_ttl, style_dp, duration_override, utt_ids)
        if noise is not None:
            self.batch_set_noise(noise)
        self.batch_run(total_step, speed, noise_seed)
        return self.batch_fetch()

    # ---- measurement -----------------------------------------------------------------------------------
    def xattn_hs_stamps_enable(self, on=True):
        self._ck(self._lib.stn_dbg_xattn_hs_enable(self._h, int(bool(on))))

    def xattn_hs_stamps(self):
        """[workgroups, 8] shader-clock stamps of the last head-split cross-attention launch (diagnostics)."""
        n = int(self._lib.stn_dbg_xattn_hs_stamps(self._h, None, 0))
        out = np.zeros((max(n, 0), 8), np.uint64)
        if n > 0:
            self._lib.stn_dbg_xattn_hs_stamps(self._h, out.ctypes.data, out.size)
        return out

    def profile_enable(self, on=True):
        self._ck(self._lib.stn_profile_enable(self._h, int(on)))

    def profile_filter(self, family=None):
        self._ck(self._lib.stn_profile_filter(self._h, family.encode() if family else None))

    def launch_log_enable(self, on=True):
        self._ck(self._lib.stn_launch_log_enable(self._h, int(bool(on))))

    def launch_log(self):
        """[(family, kernel)] of every launch since the last profile_reset (profiling on, log enabled), in dispatch order."""
        n = self._lib.stn_launch_log(self._h, None, 0)
        if n < 0:
            self._ck(int(n))
        buf = ctypes.create_string_buffer(int(n) + 1)
        self._lib.stn_launch_log(self._h, buf, int(n) + 1)
        return [tuple(l.split("\t")) for l in buf.value.decode().splitlines()]

    def profile_sample(self, every=1):
        self._ck(self._lib.stn_profile_sample(self._h, int(every)))

    def profile_reset(self):
        self._ck(self._lib.stn_profile_reset(self._h))

    def profile(self):
        """{kernel family: dict(ms, launches, flops, bytes)} measured with HIP events on the engine's stream."""
        n = self._ck(self._lib.stn_profile_count(self._h))
        out = {}
        name = ctypes.create_string_buffer(64)
        ms, fl, by, la = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        for i in range(n):
            self._ck(self._lib.stn_profile_get(self._h, i, name, 64, ctypes.byref(ms), ctypes.byref(la),
                                               ctypes.byref(fl), ctypes.byref(by)))
            out[name.value.decode()] = dict(ms=ms.value, launches=la.value, flops=fl.value, bytes=by.value)
        return out

    # ---- op-level (kernel parity tests) ----------------------------------------------------------------
    def op_gemm(self, A, W, bias=None, act=ACT_NONE, dtype=None):
        M, K = A.shape
        N = W.shape[0]
        out = np.empty((M, N), np.float32)
        _b, bptr = _opt(bias, np.float32)
        self._ck(self._lib.stn_op_gemm(self._h, self.dtype if dtype is None else _DTYPES[dtype], M, N, K,
                                       _c(A, np.float32), _c(W, np.float32), bptr, act, out))
        return out

    def op_gemm_bench(self, M, N, K, mode=0, iters=20, dtype=None):
        ms = ctypes.c_double()
        self._ck(self._lib.stn_op_gemm_bench(self._h, self.dtype if dtype is None else _DTYPES[dtype], M, N, K, mode,
                                             iters, ctypes.byref(ms)))
        return ms.value

    def op_ffn(self, xn, W1, b1, W2, b2, gamma, x, rowvec=None, row_b=None, fused=True):
        """Pointwise pair of a ConvNeXt block (K4): returns x + gamma * (W2 . GELU(W1 . xn + b1) + b2) [+ rowvec[row_b]]."""
        M, C = xn.shape
        I = W1.shape[0]
        out = np.array(x, dtype=np.float32, order="C", copy=True)
        _, b2p = _opt(b2, np.float32)
        b2a = _opt(b2, np.float32)[0]
        ga, gp = _opt(gamma, np.float32)
        rva, rvp = _opt(rowvec, np.float32)
        rba, rbp = _opt(row_b, np.int32)
        b2p = b2a.ctypes.data if b2a is not None else None
        self._ck(self._lib.stn_op_ffn(self._h, M, C, I, _c(xn, np.float32), _c(W1, np.float32), _c(b1, np.float32), _c(W2, np.float32),
                                      b2p, gp, rvp, rbp, 0 if rva is None else rva.shape[0], out, int(fused)))
        return out

    def op_ffn_bench(self, M, C, I, fused=True, iters=20):
        """fused: 0 / False two launches, 1 / True K4, 2 K4-split (partial sums only; the fold is part of op_block_bench)."""
        out = (ctypes.c_double * 5)()
        self._ck(self._lib.stn_op_ffn_bench(self._h, M, C, I, int(fused), iters, out))
        return dict(ms=out[0], first_stage=out[1], tile_loop=out[2], epilogue=out[3], workgroups=int(out[4]))

    def op_fold_dwconv_ln(self, seqlen, x, part, b2, gamma, rowvec, w, bias, g, b, k, dil):
        """fold + depthwise conv + LayerNorm on packed rows -> (x_out, y)."""
        seqlen = np.ascontiguousarray(seqlen, np.int32)
        S, M, C = part.shape
        xo = np.empty((M, C), np.float32)
        y = np.empty((M, C), np.float32)
        b2a, _ = _opt(b2, np.float32)
        ga, _ = _opt(gamma, np.float32)
        rva, _ = _opt(rowvec, np.float32)
        p = lambda a: None if a is None else a.ctypes.data
        self._ck(self._lib.stn_op_fold_dwconv_ln(self._h, len(seqlen), C, k, dil, S, seqlen, _c(x, np.float32), _c(part, np.float32), p(b2a), p(ga), p(rva),
                                                 _c(w, np.float32), _c(bias, np.float32), _c(g, np.float32), _c(b, np.float32), xo, y))
        return xo, y

    def op_block_bench(self, B, L, C, I, k, dil, mode, iters=20):
        out = (ctypes.c_double * 6)()
        self._ck(self._lib.stn_op_block_bench(self._h, B, L, C, I, k, dil, mode, iters, out))
        return dict(ms=out[0], conv_ms=out[1], fold_phase1=out[2], fold_barrier=out[3], fold_phase2=out[4], fold_span=out[5])

    def set_fused_ffn_min_rows(self, k4_rows=-1, split_rows=-1):
        self._ck(self._lib.stn_set_fused_ffn_min_rows(self._h, int(k4_rows), int(split_rows)))

    def op_gemm_phases(self, M, N, K, mode=0, dtype=None):
        out = (ctypes.c_double * 6)()
        self._ck(self._lib.stn_op_gemm_phases(self._h, self.dtype if dtype is None else _DTYPES[dtype], M, N, K, mode, out))
        return dict(first_stage=out[0], k_loop=out[1], epilogue=out[2], grid_span=out[3], entry_spread=out[4], workgroups=int(out[5]))

    def op_dwconv_ln(self, x, w, bias, g, b, dil, dtype=None, seqlen=None):
        B, L, C = x.shape
        k = w.shape[1]
        y = np.empty((B, L, C), np.float32)
        if seqlen is not None:
            self._ck(self._lib.stn_op_dwconv_ln_ragged(self._h, self.dtype if dtype is None else _DTYPES[dtype], B, L, C, k,
                                                       dil, _c(x, np.float32), _c(w, np.float32), _c(bias, np.float32),
                                                       _c(g, np.float32), _c(b, np.float32), _c(seqlen, np.int32), y))
            return y
        self._ck(self._lib.stn_op_dwconv_ln(self._h, self.dtype if dtype is None else _DTYPES[dtype], B, L, C, k, dil,
                                            _c(x, np.float32), _c(w, np.float32), _c(bias, np.float32),
                                            _c(g, np.float32), _c(b, np.float32), y))
        return y

    def op_attention(self, q, k, v, H, qlen=None, klen=None, rope_mode=-1, dtype=None):
        B, Lq, C = q.shape
        Lk = k.shape[1]
        o = np.empty((B, Lq, C), np.float32)
        _q, qp = _opt(qlen, np.int32)
        _k, kp = _opt(klen, np.int32)
        self._ck(self._lib.stn_op_attention(self._h, self.dtype if dtype is None else _DTYPES[dtype], B, Lq, Lk, H,
                                            C // H, _c(q, np.float32), _c(k, np.float32), _c(v, np.float32), qp, kp,
                                            rope_mode, o))
        return o

    def op_randn(self, seed, B, D, L, utt_ids=None, length=None):
        out = np.empty((B, D, L), np.float32)
        _u, up = _opt(utt_ids, np.int64)
        _l, lp = _opt(length, np.int32)
        self._ck(self._lib.stn_op_randn(self._h, seed, B, D, L, up, lp, out))
        return out


def pinned_array(shape, dtype):
    """A numpy array on page-locked host memory (stn_host_alloc_pinned): uploads from it are asynchronous DMA.  The memory is
    released when the array (and every view of it) is gone."""
    L = load()
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    p = L.stn_host_alloc_pinned(n)
    if not p:
        raise MemoryError("stn_host_alloc_pinned failed")
    buf = (ctypes.c_char * n).from_address(p)
    arr = np.frombuffer(buf, dtype=dt).reshape(shape)
    import weakref
    weakref.finalize(buf, L.stn_host_free_pinned, ctypes.c_void_p(p))
    return arr
