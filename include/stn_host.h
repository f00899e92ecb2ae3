/* stn_host.h — C ABI of the host-side text path (same library, libstn.so; no GPU needed by these).
 *
 * Each function replaces a piece of the reference's C++ host (paths relative to /root/reference):
 *   stn_text_preprocess    UnicodeProcessor::preprocessText           cpp/helper.cpp:52-200
 *   stn_text_to_ids        UnicodeProcessor::call                     cpp/helper.cpp:355-390
 *   stn_latent_geometry    TextToSpeech::sampleNoisyLatent (shapes)   cpp/helper.cpp:424-440,457 + getLatentMask :759-770
 *   stn_chunk_text         chunkText                                  cpp/helper.cpp:1117-1186
 *   stn_sanitize_filename  sanitizeFilename                           cpp/helper.cpp:1070-1111
 *   stn_wav_encode / stn_write_wav   writeWavFile                     cpp/helper.cpp:943-990
 *   stn_load_voice_style   loadVoiceStyle                             cpp/helper.cpp:829-897 (schema go/helper.go:87-98)
 * Strings are UTF-8, NUL-terminated.  Functions that produce text return the number of bytes needed
 * (excluding the NUL) and write at most `cap` bytes; a negative return is an STN_ERR_* code and
 * stn_host_last_error() holds the message (thread-local).
 */
#ifndef STN_HOST_H
#define STN_HOST_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* stn_host_last_error(void);

int64_t stn_text_preprocess(const char* text, const char* lang, char* out, size_t cap);

/* ids_out [B][Lt_cap] (row stride Lt_cap), lengths_out [B]; *Lt_out = max length.  If Lt_cap is too small
 * nothing is written to ids_out and *Lt_out tells the size to allocate (call twice). */
int stn_text_to_ids(const int64_t* indexer, size_t indexer_len, const char* const* texts, const char* const* langs,
                    int B, int64_t* ids_out, int Lt_cap, int32_t* lengths_out, int* Lt_out);

int stn_latent_geometry(const float* duration, int B, int sample_rate, int base_chunk_size, int chunk_compress_factor,
                        int latent_dim, int* D_out, int* L_out, int32_t* latent_lengths_out);

/* chunks are written back to back, each NUL-terminated; returns bytes needed, *n_chunks = count */
int64_t stn_chunk_text(const char* text, int max_len, char* out, size_t cap, int* n_chunks);

int64_t stn_sanitize_filename(const char* text, int max_len, char* out, size_t cap);

/* 44-byte RIFF header + int16 PCM; returns bytes needed (44 + 2n) */
/* JSON summary of an .onnx file: ir_version, graph inputs/outputs, op histogram, initializers (name, dtype, dims) */
int64_t stn_onnx_summary(const char* path, char* out, size_t cap);
/* What stn_load_dir binds when the directory has no stn_weight_map.json, without a device: the descriptor derived from the four
   graphs' weighted nodes (+ tts.json) and, per canonical tensor, the initializer it was bound to — JSON {"arch": {..},
   "tensors": {name: {"from": "...", "transpose": bool, "zeros": bool, "row0": int, "rows_total": int}}, "gelu": "erf|tanh|op|", "notes": "..."}
   (rows_total > 0: a row block of a fused q|k|v / k|v projection).  A stn_weight_map.json without a "tensors" table may state the head
   counts the graphs do not carry; with one, it replaces the walk.  Fails (STN_ERR_IO, stn_host_last_error) with
   the first node that does not fit the layout.  Stands in for Ort::Session's own graph loading, cpp/helper.cpp:784-795. */
int64_t stn_bind_graphs(const char* onnx_dir, char* out, size_t cap);
/* One canonical tensor of that binding as the engine would load it (fp32, canonical orientation: transposed / cut out of a fused
   projection as bound): returns the element count (0 for a tensor the graph does not have: zeros), copies when cap suffices. */
int64_t stn_bound_tensor(const char* onnx_dir, const char* canonical_name, float* out, size_t cap);

int64_t stn_wav_encode(const float* audio, size_t n, int sample_rate, unsigned char* out, size_t cap);
int stn_write_wav(const char* path, const float* audio, size_t n, int sample_rate);

/* Voice-style JSON files {"style_ttl": {"data": [[[..]]], "dims": [1, d1, d2]}, "style_dp": {...}} stacked along dim 0 in the
 * order given, row-major: ttl_out [n][d1][d2], dp_out [n][e1][e2].  dims6 = {n, d1, d2, n, e1, e2} (the first file's dims define
 * the layout, as in the reference).  Call with null outputs (or capacities too small: nothing is written) to learn the dims.
 * Errors ("Failed to open voice style file: <path>", data that does not match dims) come back as a negative code. */
int stn_load_voice_style(const char* const* paths, int n, float* ttl_out, size_t ttl_cap_floats, float* dp_out, size_t dp_cap_floats,
                         int64_t* dims6);

#ifdef __cplusplus
}
#endif
#endif /* STN_HOST_H */
