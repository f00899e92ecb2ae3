"""ctypes mirror of include/stn_arch.h (`struct stn_arch`) + the default descriptor."""
import ctypes

STN_MAX_VO_BLOCKS = 16

_INT_FIELDS = [
    "sample_rate", "base_chunk_size", "chunk_compress_factor", "latent_dim",
    "vocab_size", "n_style_ttl", "d_style_ttl", "n_style_dp", "d_style_dp",
    "te_dim", "te_hidden", "te_kernel", "te_conv_blocks",
    "te_attn_blocks", "te_heads", "te_ffn", "te_style_blocks", "te_out_dim",
    "dp_dim", "dp_hidden", "dp_kernel", "dp_conv_blocks", "dp_heads",
    "ve_dim", "ve_hidden", "ve_kernel", "ve_main_blocks", "ve_dilated",
    "ve_tail_blocks", "ve_heads", "ve_time_dim",
    "vo_dim", "vo_hidden", "vo_kernel", "vo_blocks", "vo_in_kernel",
]
_FLOAT_FIELDS = ["ln_eps", "rope_base", "larope_gamma", "time_scale", "head_gain"]


class StnArch(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int32) for n in _INT_FIELDS]
                + [("vo_dilations", ctypes.c_int32 * STN_MAX_VO_BLOCKS)]
                + [(n, ctypes.c_float) for n in _FLOAT_FIELDS])

    def as_dict(self):
        d = {n: getattr(self, n) for n in _INT_FIELDS + _FLOAT_FIELDS}
        d["vo_dilations"] = list(self.vo_dilations)
        return d

    @property
    def latent_channels(self):  # D = latent_dim * chunk_compress_factor
        return self.latent_dim * self.chunk_compress_factor

    @property
    def chunk_size(self):  # samples per compressed latent frame
        return self.base_chunk_size * self.chunk_compress_factor


def default_arch() -> StnArch:
    """Same values as stn_arch_default() in include/stn_arch.h (66 M parameters)."""
    a = StnArch()
    a.sample_rate, a.base_chunk_size, a.chunk_compress_factor, a.latent_dim = 44100, 512, 6, 24
    a.vocab_size, a.n_style_ttl, a.d_style_ttl, a.n_style_dp, a.d_style_dp = 512, 50, 256, 8, 16
    a.te_dim, a.te_hidden, a.te_kernel, a.te_conv_blocks = 256, 1024, 5, 6
    a.te_attn_blocks, a.te_heads, a.te_ffn, a.te_style_blocks, a.te_out_dim = 4, 4, 1024, 2, 256
    a.dp_dim, a.dp_hidden, a.dp_kernel, a.dp_conv_blocks, a.dp_heads = 128, 512, 5, 4, 2
    a.ve_dim, a.ve_hidden, a.ve_kernel, a.ve_main_blocks, a.ve_dilated = 384, 1536, 5, 4, 4
    a.ve_tail_blocks, a.ve_heads, a.ve_time_dim = 4, 4, 64
    a.vo_dim, a.vo_hidden, a.vo_kernel, a.vo_blocks, a.vo_in_kernel = 512, 2048, 7, 10, 7
    for i, d in enumerate([1, 2, 4, 1, 2, 4, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1]):
        a.vo_dilations[i] = d
    a.ln_eps, a.rope_base, a.larope_gamma, a.time_scale, a.head_gain = 1e-6, 10000.0, 10.0, 1000.0, 0.1
    return a


def tiny_arch() -> StnArch:
    """A small stack with the same topology — for fast CPU/GPU parity tests."""
    a = default_arch()
    a.n_style_ttl, a.d_style_ttl = 6, 32
    a.te_dim, a.te_hidden, a.te_conv_blocks, a.te_attn_blocks, a.te_heads, a.te_ffn = 64, 128, 2, 2, 2, 128
    a.te_style_blocks, a.te_out_dim = 1, 48
    a.dp_dim, a.dp_hidden, a.dp_conv_blocks, a.dp_heads = 32, 64, 2, 2
    (a.ve_dim, a.ve_hidden, a.ve_main_blocks, a.ve_dilated, a.ve_tail_blocks, a.ve_heads,
     a.ve_time_dim) = 96, 192, 2, 2, 1, 2, 32
    a.vo_dim, a.vo_hidden, a.vo_blocks = 64, 128, 3
    return a
