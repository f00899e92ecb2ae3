/* stn_arch.h — architecture descriptor of the four Supertonic graphs.
 *
 * The reference ships no layer definitions: all arithmetic lives in four ONNX files
 * (duration_predictor / text_encoder / vector_estimator / vocoder, loaded at
 * /root/reference/cpp/helper.cpp:784-795) that are NOT in the reference tree.  This
 * plain-int descriptor is therefore the single statement of the layer stack that the
 * engine (supertonic_amd/csrc) and the CPU oracle (oracle/stn_ref.c) both execute.
 * Defaults follow SURVEY.md Appendix C (SupertonicTTS, arXiv:2503.23108) scaled to the
 * 66 M parameters quoted at /root/reference/README.md:60.  UNVERIFIED against the real
 * graphs — every field is data, not a constant.
 *
 * Plain C, fixed-width ints, no pointers: safe to build from ctypes / cgo / JNI.
 */
#ifndef STN_ARCH_H
#define STN_ARCH_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STN_MAX_VO_BLOCKS 16

typedef struct stn_arch {
    /* tts.json fields the C++ host reads (cpp/helper.cpp:811-815) */
    int32_t sample_rate;            /* ae.sample_rate            44100 */
    int32_t base_chunk_size;        /* ae.base_chunk_size        512   */
    int32_t chunk_compress_factor;  /* ttl.chunk_compress_factor 6     */
    int32_t latent_dim;             /* ttl.latent_dim            24    */
    /* token / style geometry (unicode_indexer.json, voice-style JSON files) */
    int32_t vocab_size;             /* token ids in [0, vocab_size)            */
    int32_t n_style_ttl, d_style_ttl; /* style_ttl [B, 50, 256]                */
    int32_t n_style_dp, d_style_dp;   /* style_dp  [B, 8, 16]                  */
    /* text encoder */
    int32_t te_dim, te_hidden, te_kernel, te_conv_blocks;
    int32_t te_attn_blocks, te_heads, te_ffn, te_style_blocks, te_out_dim;
    /* duration predictor */
    int32_t dp_dim, dp_hidden, dp_kernel, dp_conv_blocks, dp_heads;
    /* vector estimator */
    int32_t ve_dim, ve_hidden, ve_kernel, ve_main_blocks, ve_dilated;
    int32_t ve_tail_blocks, ve_heads, ve_time_dim;
    /* vocoder (latent decoder) */
    int32_t vo_dim, vo_hidden, vo_kernel, vo_blocks, vo_in_kernel;
    int32_t vo_dilations[STN_MAX_VO_BLOCKS];
    /* numerics */
    float ln_eps;        /* LayerNorm epsilon                                    */
    float rope_base;     /* rotary base (10000)                                   */
    float larope_gamma;  /* length-aware RoPE scale (arXiv:2509.11084)            */
    float time_scale;    /* t in [0,1) is multiplied by this before the sinusoid  */
    float head_gain;     /* synthetic-init gain of the vocoder head               */
} stn_arch;

/* Fill `a` with the default 66 M-parameter stack. */
static inline void stn_arch_default(stn_arch* a) {
    static const int32_t dil[STN_MAX_VO_BLOCKS] = {1, 2, 4, 1, 2, 4, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
    a->sample_rate = 44100; a->base_chunk_size = 512; a->chunk_compress_factor = 6; a->latent_dim = 24;
    a->vocab_size = 512; a->n_style_ttl = 50; a->d_style_ttl = 256; a->n_style_dp = 8; a->d_style_dp = 16;
    a->te_dim = 256; a->te_hidden = 1024; a->te_kernel = 5; a->te_conv_blocks = 6;
    a->te_attn_blocks = 4; a->te_heads = 4; a->te_ffn = 1024; a->te_style_blocks = 2; a->te_out_dim = 256;
    a->dp_dim = 128; a->dp_hidden = 512; a->dp_kernel = 5; a->dp_conv_blocks = 4; a->dp_heads = 2;
    a->ve_dim = 384; a->ve_hidden = 1536; a->ve_kernel = 5; a->ve_main_blocks = 4; a->ve_dilated = 4;
    a->ve_tail_blocks = 4; a->ve_heads = 4; a->ve_time_dim = 64;
    a->vo_dim = 512; a->vo_hidden = 2048; a->vo_kernel = 7; a->vo_blocks = 10; a->vo_in_kernel = 7;
    for (int i = 0; i < STN_MAX_VO_BLOCKS; ++i) a->vo_dilations[i] = dil[i];
    a->ln_eps = 1e-6f; a->rope_base = 10000.0f; a->larope_gamma = 10.0f; a->time_scale = 1000.0f;
    a->head_gain = 0.1f;
}

#ifdef __cplusplus
}
#endif
#endif /* STN_ARCH_H */
