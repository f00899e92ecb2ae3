/* stn.h — C ABI of the MI355X-native Supertonic synthesis engine (libstn.so).
 *
 * This is the drop-in boundary for the reference's hot path.  Each entry point names the reference
 * interface it replaces; all paths are relative to /root/reference.
 *
 *   reference (ONNX Runtime C++ API)                      here
 *   ------------------------------------------------      -------------------------------------------
 *   Ort::Session ctor x4, loadOnnxAll                     stn_create + stn_load_dir / stn_load_synthetic
 *     cpp/helper.cpp:776-795, loadCfgs :801-818
 *   dp_ort_->Run          cpp/helper.cpp:512-526          stn_duration
 *   text_enc_ort_->Run    cpp/helper.cpp:545-556          stn_text_enc
 *   vector_est_ort_->Run  cpp/helper.cpp:620-658          stn_vector_est   (one Euler step per call)
 *   vocoder_ort_->Run     cpp/helper.cpp:662-679          stn_vocoder
 *   TextToSpeech::_infer  cpp/helper.cpp:469-683          stn_batch_upload + stn_batch_run + stn_batch_fetch
 *   sampleNoisyLatent     cpp/helper.cpp:424-467          inside stn_batch_run (Philox noise, or injected)
 *
 * Conventions: plain C, no exceptions cross the boundary.  Every function returns STN_OK (0) or a
 * negative error code; the message is available from stn_last_error().  Tensors are contiguous,
 * row-major, caller-owned HOST buffers with the names / dtypes / shapes of the ONNX graph I/O
 * (text_ids int64 [B,Lt]; masks float32 [B,1,L]; step counters float32 [B]; everything else float32).
 * One handle = one GPU + one HIP stream; calls on one handle must be serialised by the caller; distinct
 * handles are independent (one per GPU).  There is NO CPU fallback: without a HIP device stn_create fails.
 */
#ifndef STN_H
#define STN_H
#include <stddef.h>
#include <stdint.h>

#include "stn_arch.h"

#ifdef __cplusplus
extern "C" {
#endif

#define STN_OK 0
#define STN_ERR_INVALID (-1)      /* bad argument / shape                         */
#define STN_ERR_DEVICE (-2)       /* HIP failure or no device                     */
#define STN_ERR_STATE (-3)        /* call order (no model loaded, no batch, ...)  */
#define STN_ERR_UNSUPPORTED (-4)  /* feature not available in this build          */
#define STN_ERR_IO (-5)           /* file missing / unreadable / malformed        */

#define STN_DTYPE_F32 0   /* fp32 operands, exact fp32 MFMA                        */
#define STN_DTYPE_BF16 1  /* bf16 GEMM operands, fp32 accumulate + fp32 residual   */
#define STN_DTYPE_F16 2   /* IEEE half GEMM operands / activations (v_mfma_f32_32x32x16_f16), fp32 accumulate + fp32 residual:
                           * BASELINE config 5 ("fp16 MFMA linears"); 11 significant bits instead of bf16's 8, range 6.5e4 */

typedef struct stn_handle stn_handle;

typedef struct stn_config {
    int32_t device;  /* HIP device ordinal                                         */
    int32_t dtype;   /* STN_DTYPE_*                                                */
} stn_config;

/* ---- lifetime / model load -------------------------------------------------------------------- */
int stn_create(const stn_config* cfg, stn_handle** out);
int stn_destroy(stn_handle* h);
/* message of the last failure on this handle (or of the last failed stn_create when h == NULL) */
const char* stn_last_error(const stn_handle* h);
/* The reference's asset directory (cpp/helper.cpp:784-823): tts.json + unicode_indexer.json + the four .onnx graphs, nothing else.
 * tts.json gives the descriptor's config fields; the graphs are read by a built-in protobuf reader and their NODES are walked in
 * graph order: the nodes that carry weights (depthwise Conv, pointwise Conv / MatMul+Add / Gemm, LayerNormalization, layer-scale
 * Mul, Gather) are parsed against the engine's layout (embedding / ConvNeXt blocks / attention blocks / projections), which yields
 * the rest of the descriptor (widths, depths, kernel sizes, dilations, head counts from the Reshape constants) and binds every
 * initializer to its canonical tensor by position and role — never by name (csrc/host/graph_bind.hpp; stn_bind_graphs in
 * stn_host.h shows the result without a device).  Graph input/output names must be the ones the hosts use (cpp/helper.cpp:512-513,
 * 545-546, 620-623, 663-664); tts.json must agree with the shapes.  A graph that is not this layout fails with the first node that
 * does not fit, what the layout needs there, and the descriptor derived so far.
 * Optional `<onnx_dir>/stn_weight_map.json` overrides the walk with an explicit table
 *   {"arch": {<stn_arch field>: int, ...}, "tensors": {"<engine tensor>": {"file": "vocoder.onnx", "name": "<initializer>",
 *    "transpose": false}, ...}}   ("transpose" may be omitted: it is inferred from the stored dims, which are checked either way).
 * STN_ERR_IO: a file is missing/unreadable/malformed ("Failed to open ..." as cpp/helper.cpp:805) or does not bind;
 * STN_ERR_INVALID: the derived descriptor is outside what the kernels support (message names the field).
 * After a successful manifest-less load stn_last_error holds notes (e.g. that the graphs spell GELU with Tanh).  A head count the graphs do
 * not carry (no [batch, length, heads, head_dim] Reshape constant) is an ERROR, not a default: state it in a stn_weight_map.json that has only
 * {"arch": {"te_heads": n, "dp_heads": n, "ve_heads": n}} (no "tensors" table: the graphs are still walked).
 * Accepted spellings of the layout: LayerNormalization as one node or decomposed (ReduceMean / Sub / Pow / ReduceMean / Add / Sqrt / Div / Mul /
 * Add), q / k / v projections separate or fused (3C or 2C rows + Split), pointwise convolutions as Conv k=1, MatMul(+Add) with Transposes around it
 * or Gemm, GELU as a node or written with Erf / Tanh, the vocoder's wave head as a projection or as a one-channel ConvTranspose with stride ==
 * kernel == base_chunk_size (an overlapping transposed convolution is another head: refused with the node named).  VERIFIED ON GRAPHS THE TESTS EMIT (tests/onnx_graphs.py), not on the published files, which
 * are not available offline: for real assets the explicit stn_weight_map.json below is the documented path, and the walk says where a graph departs
 * from the layout. */
int stn_load_dir(stn_handle* h, const char* onnx_dir);
/* '\n'-separated names of every canonical tensor the descriptor implies (what a manifest must map); returns bytes needed */
int stn_tensor_names(stn_handle* h, const stn_arch* arch, char* out, size_t cap);
/* descriptor-driven deterministic weights (no asset files needed) */
int stn_load_synthetic(stn_handle* h, const stn_arch* arch, uint64_t seed);
int stn_get_arch(const stn_handle* h, stn_arch* out);
int64_t stn_param_count(const stn_handle* h);

/* ---- the four former Run sites (host pointers in, host pointers out) ------------------------------ */
int stn_duration(stn_handle* h, int B, int Lt, const int64_t* text_ids, const float* style_dp /*[B,e1,e2]*/,
                 const float* text_mask /*[B,1,Lt]*/, float* duration /*[B] seconds*/);
int stn_text_enc(stn_handle* h, int B, int Lt, const int64_t* text_ids, const float* style_ttl /*[B,d1,d2]*/,
                 const float* text_mask, float* text_emb /*[B,Ce,Lt]*/);
int stn_vector_est(stn_handle* h, int B, int L, int Lt, const float* noisy_latent /*[B,D,L]*/,
                   const float* text_emb /*[B,Ce,Lt]*/, const float* style_ttl, const float* text_mask,
                   const float* latent_mask /*[B,1,L]*/, const float* total_step /*[B]*/,
                   const float* current_step /*[B], 0-based*/, float* denoised_latent /*[B,D,L]*/);
int stn_vocoder(stn_handle* h, int B, int L, const float* latent /*[B,D,L]*/, float* wav /*[B, L*cs]*/);

/* ---- fused synthesis: everything stays in HBM between stages ---------------------------------------
 * upload: copies the batch to the GPU (text_ids, text_mask, styles; optional duration override in
 *         seconds BEFORE the /speed division; optional utterance ids that key the noise generator so a
 *         sharded batch draws the same noise as an unsharded one).
 * run:    DP -> /speed -> text encoder -> noise -> total_step x estimator -> vocoder, all on the handle's
 *         stream; returns after ENQUEUE (asynchronous) except for one tiny device->host read of the
 *         durations when no override is given.  Call stn_sync or stn_batch_fetch to wait.
 * fetch:  waits and copies out wav [B, L*cs] and duration [B] (after /speed, as the reference returns). */
int stn_batch_upload(stn_handle* h, int B, int Lt, const int64_t* text_ids, const float* text_mask,
                     const float* style_ttl, const float* style_dp, const float* duration_override_or_null,
                     const int64_t* utt_ids_or_null);
int stn_batch_set_noise(stn_handle* h, const float* noise /*[B,D,L]*/, int L);
int stn_batch_run(stn_handle* h, int total_step, float speed, uint64_t noise_seed);
/* hipGraph replay of the post-duration pipeline (default on): a shape is captured the second time it is run and replayed
 * afterwards.  Up to 8 captured shapes are kept (least recently used out first), so callers that alternate a few shapes — the
 * reference's call() chunk loop and n_test loop, cpp/helper.cpp:697-719, cpp/example_onnx.cpp:88 — replay all of them; loading
 * weights on the handle drops every captured graph.  stn_graph_replays counts replays, stn_graphs_cached the graphs held
 * (tests / diagnostics). */
int stn_set_graph_mode(stn_handle* h, int on);
int64_t stn_graphs_cached(const stn_handle* h);
/* Vocoder treatment of the padding in stn_batch_run.  0 (default) = the reference's batched Run (cpp/helper.cpp:668-679):
 * all L*ccf frames of every utterance are decoded, the padding being zero latent.  1 = length-aware: each utterance's frames
 * end at its own latent length, so wav[b, :len_b] is what a batch-of-one synthesis of utterance b gives — the mode in which
 * the chunks of a long text (TextToSpeech::call, cpp/helper.cpp:685-722, one _infer per chunk) run as ONE batch. */
int stn_set_vocoder_mode(stn_handle* h, int length_aware);
/* Row layout of the vector estimator inside stn_batch_run.  1 (default) = packed: the estimator's activations hold only the
 * latent frames each utterance owns (sum of lengths rows), 0 = padded [b*L + t] rows with the padding masked to zero after
 * every block.  The masked stages are row-independent, so both give the same latent; packed does no work on padding. */
int stn_set_row_layout(stn_handle* h, int packed);
/* Shape buckets for the graph cache (default off).  A captured pipeline has every size baked in: B, Lt, L and the packed row counts.
 * With buckets on, Lt, L and the row counts are rounded UP to bucket boundaries (4-8 per octave, at most 25 % padding), so requests of
 * unlike lengths that fall into the same buckets replay ONE captured graph instead of capturing one each — the case of a service
 * that merges requests (/root/reference/py/service.py:79-136) and of the call() chunk loop (/root/reference/cpp/helper.cpp:697-719).
 * The utterances' own lengths stay exact: every utterance's frames and samples are bit-identical to the unbucketed run.  What the
 * caller sees: stn_batch_dims reports the bucketed L (W = L * chunk samples) and fetched rows have that length (zeros / the dense
 * vocoder's padding response behind an utterance's own samples, as before); B stays exact; injected noise keeps L exact. */
int stn_set_shape_buckets(stn_handle* h, int on);
/* Measurement aid.  stn_batch_upload with a duration override skips the one device->host read of a synthesis (the predicted
 * durations, which size every later buffer): with always != 0 the read and the wait for it are performed anyway, so a timed run has
 * the critical path of a predicted-duration run (duration predictor -> read -> rest) on the controlled shapes of a forced one. */
int stn_set_duration_read(stn_handle* h, int always);
/* GELU form of the loaded model: 0 = erf (default), 1 = the tanh approximation 0.5 x (1 + tanh(sqrt(2/pi)(x + 0.044715 x^3))).
 * stn_load_dir sets it from how the graphs spell the activation (Gelu / Erf: 0; Tanh inside the GELU pattern or Gelu approximate="tanh":
 * 1); a synthetic-weight engine that should compute the tanh form sets it here.  fp32 and f16 engines follow it exactly (the fused K4
 * kernels of the 16-bit modes compute the exp2 = tanh form either way, bf16 stores the tanh-form shortcut: DESIGN.md 5d). */
int stn_set_gelu_form(stn_handle* h, int tanh_form);
int stn_get_gelu_form(const stn_handle* h);
/* Cross-attention blocks of the vector estimator: non-zero (default) = head-split: fold_ln (or LayerNorm) + ONE launch per block that
 * computes, per (utterance pair, head), the q projection, its rotation, the attention and the head's share of the output projection
 * and leaves it as a 16-bit per-head partial sum which the next ConvNeXt block's fold adds to the residual stream in head order
 * (kernels_xattn_hs.hip; 16-bit modes, packed rows with K4-split active, <= 256 frames per utterance, contexts of <= 128 keys: other
 * shapes take the four launches); 0 = four launches (LayerNorm, q projection, attention, output projection + residual).  Same result up
 * to the rounding of the per-head partial sums (tests/test_gpu_xattn.py).  (Rounds 1-3 had per-utterance one- and two-launch forms
 * here; they measured at parity or slower and were retired when the head-split form replaced them.) */
int stn_set_fused_xattn(stn_handle* h, int on);
/* K4 — the pointwise pair of a ConvNeXt block (pw1 -> GELU -> pw2 -> layer scale + residual) as ONE launch whose 4C-wide hidden
 * activation never leaves the registers (bf16 and f16 engines, block widths 384 / 512, batches of >= 18432 rows — below that a workgroup per 128 rows leaves most of the chip
 * idle while each still streams both weight matrices; other shapes keep the two GEMM launches).
 * Bit mask over the stages: 1 = vocoder, 2 = vector estimator, 4 = text encoder / duration predictor, 8 = the estimator's blocks as
 * K4-split (see stn_set_fused_ffn_min_rows); 0 = never.  The default (9) is the set of stages where a form measured faster on
 * MI355X.
 * WHAT IS AND IS NOT BIT-IDENTICAL.  Which kernel a block runs is decided by the launch's row count (these thresholds; the K4-split of
 * the estimator's blocks is 12 ways up to 1536 packed rows, 8 ways up to 4096 and 4 ways beyond; also split-K for small exact-fp32 GEMMs and the attention
 * grid shape), so an utterance synthesized alone, a small shard of a batch (e.g. 16 utterances per GPU of a strong-scaling run: below
 * 1536 rows) and the same utterance inside a large batch may run different kernels: they agree to rounding (K4 vs two launches: fp32 summation order and, in f16, the exp2- vs
 * erf-form GELU; K4-split: 16-bit partial sums), with identical predicted durations to 1e-5 and identical latent lengths
 * (tests/test_gpu_batch_invariance.py holds recorded bounds).  Within one kernel regime a row's result does not depend on the
 * number of rows, their order or their position in the launch: packed vs trimmed vs dense vocoder rows, graph replay vs eager and
 * sharded vs unsharded batches of the same regime are bit-identical (tests/test_gpu_packed.py, tests/test_gpu_ffn.py). */
int stn_set_fused_ffn(stn_handle* h, int stage_mask);
/* Row thresholds of the two fused forms (negative: leave unchanged): K4 (stage mask bits 1 / 2 / 4) is taken from k4_rows rows on,
 * K4-split (bit 8: the estimator's blocks with the hidden dimension cut over four workgroups per 128-row slab and 16-bit partial
 * sums folded by the next reader of the residual stream) from split_rows rows on.  Below a threshold the block runs as two tiled
 * GEMM launches; the forms agree to rounding, not bit for bit (tests/test_gpu_ffn.py, tests/test_gpu_batch_invariance.py). */
int stn_set_fused_ffn_min_rows(stn_handle* h, int64_t k4_rows, int64_t split_rows);
/* rows the vector estimator worked on in the last stn_batch_run: sum of the latent lengths (packed) or B*L (padded) */
int64_t stn_batch_ve_rows(const stn_handle* h);
/* frames the vocoder computed in the last stn_batch_run: B*L*ccf, or fewer when the position-independent part of the padding
 * was filled from the model's cached zero-latent response (bit-identical result; bf16 and f16 engines, packed row layout) */
int64_t stn_batch_vo_rows(const stn_handle* h);
int64_t stn_graph_replays(const stn_handle* h);
int stn_batch_dims(const stn_handle* h, int* B, int* L, int64_t* wav_len_per_utt);
int stn_batch_fetch(stn_handle* h, float* wav, size_t wav_capacity_floats, float* duration);
/* same, as 16-bit PCM converted on the GPU exactly as writeWavFile does (clamp to [-1,1], *32767, truncation;
 * cpp/helper.cpp:986-987): half the device->host bytes */
int stn_batch_fetch_pcm16(stn_handle* h, int16_t* pcm, size_t capacity_samples, float* duration);
/* the same, pipelined over two slots: _begin converts to PCM on the GPU and starts the device->host copy on a second stream,
 * so that the copy of batch i overlaps stn_batch_upload / stn_batch_run of batch i+1; _end waits for the slot's copy and hands
 * out the handle's pinned host buffer ([B, W] int16, valid until the slot's next _begin) with the batch's durations.  This is
 * the host-to-host path of _infer's contract (host inputs in, host waveform out: cpp/helper.cpp:674-682) at full overlap. */
int stn_batch_fetch_pcm16_begin(stn_handle* h, int slot /* 0 or 1 */);
int stn_batch_fetch_pcm16_end(stn_handle* h, int slot, const int16_t** pcm, size_t* n_samples, float* duration_or_null);
/* the batch a slot holds (fixed at its _begin; the resident batch may have changed since): utterances, samples per utterance;
 * `duration_or_null` of _end takes B floats */
int stn_batch_fetch_slot_dims(stn_handle* h, int slot, int* B, int64_t* samples_per_utt);
/* page-locked host memory for the caller's input buffers (ids, masks, styles): uploads from it are asynchronous DMA instead of a
 * staged copy.  NULL on failure. */
void* stn_host_alloc_pinned(size_t bytes);
void stn_host_free_pinned(void* p);
int stn_batch_fetch_latent(stn_handle* h, float* latent /*[B,D,L]*/);
/* device pointer of the finished waveform [B, L*cs] float32 (valid until the next upload/run) */
int stn_batch_wav_device_ptr(const stn_handle* h, void** ptr);
int stn_sync(stn_handle* h);
/* enqueue on a caller-owned HIP stream (hipStream_t passed as void*; NULL = the handle's own stream), e.g. the
 * framework stream that also carries the RCCL gather of the finished waveforms */
int stn_set_stream(stn_handle* h, void* hip_stream);
/* device->device copy of the finished waveform rows [B][W] into dst (row stride dst_stride floats >= W), enqueued
 * on the handle's stream */
int stn_batch_copy_wav_device(stn_handle* h, void* dst_device, int64_t dst_stride);
/* same as 16-bit PCM (the conversion of writeWavFile, cpp/helper.cpp:986-987), e.g. straight into an RCCL gather payload:
 * half the bytes over xGMI; dst_stride in samples */
int stn_batch_copy_pcm16_device(stn_handle* h, void* dst_device, int64_t dst_stride);

/* ---- measurement: HIP-event timing of kernel families on the engine's own stream ------------------- */
int stn_profile_enable(stn_handle* h, int on);
int stn_profile_reset(stn_handle* h);
/* time only one kernel family ("stage.kernel", e.g. "vo.gemm_pw1_gelu"); NULL or "" = all families */
int stn_profile_filter(stn_handle* h, const char* family_or_null);
/* time only every n-th matching launch (n >= 1): a launch that carries events does not overlap its neighbours on the stream
   (~4 us each), so a timed region samples its dominant family instead of fencing every launch of it */
int stn_profile_sample(stn_handle* h, int every);
/* Launch log for profiler runs: while on (and profiling enabled) every kernel launch of this thread's engine calls is recorded
   as "family\tkernel\n" in dispatch order (family "-" outside a timed family); stn_launch_log returns the bytes needed and fills
   `out` when it fits.  tools/pmc_families.py aligns rocprofv3's per-dispatch rows with it.  Cleared by stn_profile_reset. */
int stn_launch_log_enable(stn_handle* h, int on);
int64_t stn_launch_log(stn_handle* h, char* out, size_t cap);
/* number of kernel families seen; then per index: name, total ms, launches, algorithmic flops and bytes */
int stn_profile_count(stn_handle* h);
int stn_profile_get(stn_handle* h, int idx, char* name, size_t name_cap, double* total_ms, int64_t* launches,
                    double* flops, double* bytes);

/* diagnostics (tools/xattn_hs_phases.py): with on != 0 every head-split cross-attention launch (stn_set_fused_xattn(h, 3), eager runs
   only) writes 8 shader-clock stamps per workgroup — entry, Wq in LDS, q projection done, K/V in LDS, attention done, Wo in LDS, end,
   and the workgroup's row-tile count — into one buffer; stn_dbg_xattn_hs_stamps copies the LAST launch's stamps (up to cap values)
   and returns how many workgroups that launch had. */
int stn_dbg_xattn_hs_enable(stn_handle* h, int on);
int64_t stn_dbg_xattn_hs_stamps(stn_handle* h, unsigned long long* out, size_t cap);
/* diagnostics (no device needed): the run length — frames per workgroup, 0 = the default of 32 — the estimator's fold + conv + LayerNorm kernel takes for a
   batch of these latent lengths on a device of `n_cu` compute units (one 1024-thread workgroup fills a CU: the grid is kept within ONE round of workgroups
   where a run of 40 or 48 frames achieves that; the results do not depend on the choice).  < 0: STN_ERR_INVALID. */
int stn_dbg_fold_run_frames(const int32_t* latent_lengths, int B, int n_cu);

/* ---- op-level entry points used by the kernel parity tests (host pointers) -------------------------- */
int stn_op_gemm(stn_handle* h, int dtype, int M, int N, int K, const float* A /*[M,K]*/, const float* W /*[N,K]*/,
                const float* bias_or_null, int act /*0 none,1 gelu,2 silu*/, float* out /*[M,N]*/);
/* device-resident timing of one GEMM shape on random operands; mode 0 = bias+GELU store, 1 = residual epilogue */
int stn_op_gemm_bench(stn_handle* h, int dtype, int M, int N, int K, int mode, int iters, double* avg_ms);
/* diagnostics: shader-clock phase stamps of ONE launch of the tiled GEMM kernel on this shape (mode as above).  out6 = mean
 * cycles to the first landed stage, in the K loop, in the epilogue; span of the whole grid; spread of workgroup entry times;
 * number of workgroups.  Cycles of the s_memtime counter (100 MHz on gfx950: 1 tick = 10 ns). */
int stn_op_gemm_phases(stn_handle* h, int dtype, int M, int N, int K, int mode, double* out6);
int stn_op_dwconv_ln(stn_handle* h, int dtype, int B, int L, int C, int k, int dil, const float* x,
                     const float* w /*[C,k]*/, const float* bias, const float* ln_g, const float* ln_b, float* y);
/* same with per-sequence valid lengths (taps at t >= seqlen[b] read as zero, rows t >= seqlen[b] come back as zeros) */
int stn_op_dwconv_ln_ragged(stn_handle* h, int dtype, int B, int L, int C, int k, int dil, const float* x,
                            const float* w /*[C,k]*/, const float* bias, const float* ln_g, const float* ln_b,
                            const int32_t* seqlen /*[B], 0..L*/, float* y);
/* rope_mode: -1 none, 0 RoPE on the position index, 1 length-aware RoPE; OR-ing 0x100 makes the engine rotate the keys in
 * a separate pass first (the way the vector estimator's step-invariant text keys are handled) — same result */
int stn_op_attention(stn_handle* h, int dtype, int B, int Lq, int Lk, int H, int dh, const float* q, const float* k,
                     const float* v, const int32_t* qlen_or_null, const int32_t* klen_or_null, int rope_mode, float* o);
/* the pointwise pair of a ConvNeXt block on host operands (16-bit engines): x <- x + gamma * (W2 . GELU(W1 . xn + b1) + b2)
 * [+ rowvec[row_b[m]]], W1 [I,C], W2 [C,I], x [M,C] in place.  fused = 1: the K4 kernel, 0: the two tiled GEMM launches,
 * 2: K4-split (16-bit partial sums of the hidden quarters) followed by the fold. */
int stn_op_ffn(stn_handle* h, int M, int C, int I, const float* xn /*[M,C]*/, const float* W1, const float* b1, const float* W2,
               const float* b2_or_null, const float* gamma_or_null, const float* rowvec_or_null /*[nseq,C]*/,
               const int32_t* row_b_or_null /*[M]*/, int nseq, float* x, int fused);
/* timing of the same on random device-resident operands.  out5: avg ms per call; fused only: mean shader-clock cycles per
 * workgroup until the first stage landed / in the tile loop / in the epilogue, and the number of workgroups */
int stn_op_ffn_bench(stn_handle* h, int M, int C, int I, int fused, int iters, double* out5);
/* fold + depthwise conv + LayerNorm on packed rows (16-bit engines): sequence b owns seqlen[b] consecutive rows, M = their sum;
 * part [S,M,C] (fp32, rounded to the engine's 16-bit format first).  x_out [M,C] = x + gamma * (sum_s part[s] + b2) + rowvec[b],
 * y [M,C] = LayerNorm(dwconv_{k,dil}(x_out)) (fp32 copy of the 16-bit output). */
int stn_op_fold_dwconv_ln(stn_handle* h, int B, int C, int k, int dil, int S, const int32_t* seqlen, const float* x, const float* part,
                          const float* b2_or_null, const float* gamma_or_null, const float* rowvec_or_null /*[B,C]*/, const float* w /*[C,k]*/,
                          const float* bias, const float* ln_g, const float* ln_b, float* x_out, float* y);
/* timing of one estimator-style block on B packed sequences of L frames, random device-resident operands: mode 0 = dwconv_ln + pw1 +
 * pw2, mode 2 = fold_dwconv_ln + K4-split.  out6: avg ms per block, avg ms of its conv kernel alone; mode 2: mean shader-clock cycles per
 * fold_dwconv_ln workgroup in phase 1 / at the hand-over barrier / in phase 2, and the launch's span (first entry to last exit) */
int stn_op_block_bench(stn_handle* h, int B, int L, int C, int I, int k, int dil, int mode, int iters, double* out6);
int stn_op_randn(stn_handle* h, uint64_t seed, int B, int D, int L, const int64_t* utt_ids_or_null,
                 const int32_t* len_or_null, float* out);

const char* stn_version(void);
/* HIP runtime versions: the one libstn.so was compiled against and the one it runs on (they differ when the process loaded
 * PyTorch-ROCm's bundled runtime first); no device needed */
int stn_hip_versions(int* built, int* runtime);
/* Visible HIP devices (< 0: STN_ERR_DEVICE), and a device-wide synchronize (hipDeviceSynchronize on `device`): what a host that does not
 * link a HIP runtime itself needs around a timed region (bench.py --gpus 1 runs without PyTorch, on the runtime this library ships against). */
int stn_device_count(void);
int stn_device_sync(int device);
/* forms of the pointwise pair the kernels offer for a block shape (no device needed): 0 = two tiled launches only, 1 = K4,
 * 2 = K4 and K4-split.  Counts the LDS a workgroup needs (ring + biases <= 160 KiB), so a descriptor that loads never selects a
 * kernel that cannot launch. */
int stn_ffn_fused_forms(int dtype, int C, int I);

#ifdef __cplusplus
}
#endif
#endif /* STN_H */
