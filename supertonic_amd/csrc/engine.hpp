// engine.hpp — MI355X-native executor of the four Supertonic graphs.
//
// Replaces what the reference does with four `Ort::Session`s (/root/reference/cpp/helper.cpp:776-795) and
// their `Run` calls (:519, :552, :643, :668).  One Engine = one GPU + one HIP stream; all stages are
// enqueued on that stream with no host round trip except the single read of the predicted durations that
// sizes the latent (cpp/helper.cpp:430-438 needs max(duration) on the host too).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/stn_arch.h"
#include "kernels.hpp"

namespace stn {

#define STN_HIP(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess)                                                                                  \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr);       \
    } while (0)

struct DevTensor {
    float* f32 = nullptr;      // canonical fp32 copy (layout noted per tensor)
    uint16_t* bf16 = nullptr;  // 16-bit copy for GEMM operands (matrices only) in the engine's format: bf16, or IEEE half for F16 engines
    int rows = 0, cols = 0;
    const void* as(int dt) const { return is_half(dt) ? static_cast<const void*>(bf16) : static_cast<const void*>(f32); }
};

struct Linear { DevTensor w; const float* b = nullptr; int N = 0, K = 0; };
struct LNorm { const float* g = nullptr; const float* b = nullptr; };
struct ConvNeXt { const float* dw_t = nullptr; const float* dw_b = nullptr; LNorm ln; Linear pw1, pw2; const float* gamma = nullptr; };
struct Attn { LNorm ln; Linear q, kv, qkv, o; };  // kv = [Wk; Wv] rows, qkv = [Wq; Wk; Wv] rows

// Grow-only chunked device workspace.  Stage code takes mark()/release() pairs; all work is stream-ordered,
// chunks are never freed before the Engine dies, so pointers handed out stay valid while kernels run.
// New chunks are hipMalloc'ed only the first time a shape needs them (warm-up), never afterwards.
class Arena {
   public:
    struct Mark { size_t chunk, off; };
    Arena() = default;
    Arena(const Arena&) = delete;             // owns device memory: a copy would free it twice
    Arena& operator=(const Arena&) = delete;
    ~Arena();
    void swap(Arena& o) { chunks_.swap(o.chunks_); std::swap(cur_, o.cur_); std::swap(off_, o.off_); }
    void* alloc(size_t bytes);
    Mark mark() const { return {cur_, off_}; }
    void release(Mark m) { cur_ = m.chunk; off_ = m.off; }
    void reset() { cur_ = 0; off_ = 0; }
    size_t capacity() const;

   private:
    struct Chunk { char* p; size_t cap; };
    std::vector<Chunk> chunks_;
    size_t cur_ = 0, off_ = 0;
};

struct KernelStat { double ms = 0; long launches = 0; double flops = 0; double bytes = 0; };

class Engine {
   public:
    Engine(int device, int dtype);
    ~Engine();
    Engine(const Engine&) = delete;

    void load_synthetic(const stn_arch& a, uint64_t seed);
    // canonical tensors by name from any source (ONNX initializers through a manifest, ...): fn(name, rows, cols) must return
    // rows*cols floats in the canonical layout (Linear [N][K]; depthwise [C][k]; vocoder input conv [Cout][Cin][k]; 1-D as [1][n])
    using TensorSource = std::function<std::vector<float>(const std::string& name, int rows, int cols)>;
    void load_tensors(const stn_arch& a, const TensorSource& src);
    std::vector<std::string> tensor_names(const stn_arch& a);  // every canonical tensor the descriptor implies (no device work)
    bool loaded() const { return loaded_; }
    const stn_arch& arch() const { return a_; }
    int64_t param_count() const { return params_; }
    int dtype() const { return dt_; }
    hipStream_t stream() const { return s_; }
    void sync() { STN_HIP(hipStreamSynchronize(s_)); if (dp_s_) STN_HIP(hipStreamSynchronize(dp_s_)); if (te_s_) STN_HIP(hipStreamSynchronize(te_s_)); }
    // run on a caller-owned stream (e.g. torch's current stream, so RCCL ops order after the engine's kernels);
    // nullptr returns to the engine's own stream
    void set_stream(hipStream_t s);

    // ---- device-level stages (all pointers device, enqueued on stream()) -------------------------
    // lengths are int32 [B] on device.
    struct Ragged;  // packed rows (defined below)
    // trg (optional): the text positions as packed rows (sum of tlen rows, off[b] = first row of utterance b)
    void duration_dev(int B, int Lt, const int64_t* ids, const float* style_dp, const int* tlen, float* dur, const Ragged* trg = nullptr);
    // emits text_emb as NCL fp32 [B,Ce,Lt] (if ncl) and/or as rows [B*Lt][Ce] in the act dtype (if rows)
    void text_enc_dev(int B, int Lt, const int64_t* ids, const float* style_ttl, const int* tlen, float* ncl, void* rows,
                      const Ragged* trg = nullptr);
    struct VeCtx { void* text_kv = nullptr; void* style_kv = nullptr; int Lt = 0; const int* text_off = nullptr; };  // step-invariant K/V
    // tlen: text lengths (the text keys are rotated here, once, with their length-aware positions)
    // defer_text: allocate the text K/V but leave them to ve_text_kv_dev (batch_run computes them behind the text-row hand-over)
    VeCtx ve_prepare_dev(int B, int Lt, const void* text_rows, const float* style_ttl, const int* tlen, const Ragged* trg = nullptr, bool defer_text = false);
    void ve_text_kv_dev(const VeCtx& c, int B, int Lt, const void* text_rows, const int* tlen, const Ragged* trg);
    // time conditioning of `rows` (= B x steps) (current, total) pairs -> tb [rows][main_blocks * C] (fp32, arena)
    float* ve_time_cond_dev(int rows, const float* total_step, const float* current_step, float* tb_out = nullptr /* else: from the arena */);
    // The resident batch's time conditioning depends on (total_step, B, weights) and on nothing the caller uploads — every utterance of a run has the same
    // step counters — so it is computed once per such triple into a persistent buffer, eagerly and outside the captured pipeline, instead of by every
    // synthesis (time embedding + three exact-fp32 GEMMs + the step counters: ~75 us of a 10.8 ms batch, five launches)
    struct TimeCond { int steps = 0, B = 0; uint64_t wgen = ~0ull; float* buf = nullptr; size_t cap = 0; float *tot = nullptr, *cur = nullptr, *dt = nullptr, *tb = nullptr; };
    TimeCond tcond_;
    void ensure_time_cond(int total_step, int B);
    // tb: rows of this step's time conditioning ([B][main_blocks*C]); nullptr -> computed here from the step counters
    // Packed ("ragged") latent rows: utterance b owns rows off[b] .. off[b] + llen[b] and no padding rows exist; `rows` is
    // their total.  The masked stages are row-independent, so this is an exact optimisation of the padded [b*L + t] layout.
    struct Ragged { const int* off = nullptr; const int* row_b = nullptr; int rows = 0;
                    const int* hs_pairs = nullptr;  // (latent rows) which two utterances share a workgroup of the head-split cross-attention
                    int fold_run = 0; };            // (latent rows) run length of fold_dwconv_ln's workgroups for these lengths (fold_run_frames; 0: default)
    void ve_step_dev(int B, int L, const VeCtx& c, const float* noisy, const int* tlen, const int* llen,
                     const float* total_step, const float* current_step, float* denoised, const float* tb = nullptr,
                     const Ragged* rg = nullptr, const float* dt = nullptr /* 1/total_step per utterance, if precomputed */,
                     void* z_rows = nullptr /* [rows][D padded to 64] act, columns >= D zero: persistent across the steps of one synthesis */,
                     bool z_ready = false /* z_rows already holds `noisy` as rows (the step before wrote it) */,
                     bool z_next = false /* write `denoised` into z_rows as rows for the step after */);
    // vlen (optional, device [B]): valid vocoder frames per utterance — the length-aware mode (see set_vocoder_mode)
    // vrows > 0 (with vlen): run the vocoder on packed rows — vrows = sum of vlen — and unpack into the padded wav at the end
    // valid (with vlen, vrows): the exact trimmed DENSE mode — rows are computed on vlen[b] frames, exact below valid[b], and
    // the rest of each row is the cached zero-latent response (quiet chunk + edge tail)
    void vocoder_dev(int B, int L, const float* latent, float* wav, const int* vlen = nullptr, int vrows = 0,
                     const int* valid = nullptr);
    int vocoder_receptive_field() const;  // frames on either side that one output frame depends on
    void prepare_xattn_weights();         // fragment-ordered copies of the estimator's cross-attention Wq / Wo (kernels_xattn_hs.hip)
    std::unordered_map<const void*, const void*> frag_w_;  // row-major 16-bit matrix -> its fragment-ordered copy
    // diagnostics: stamps of the head-split cross-attention launches (stn_dbg_xattn_hs_*)
    void hs_stamps_enable(bool on);
    int64_t hs_stamps_fetch(unsigned long long* out, size_t cap);
    unsigned long long* hs_ts_ = nullptr;  // 8 values per workgroup, HS_TS_WG workgroups
    int hs_ts_wgs_ = 0;                    // workgroups of the last stamped launch
    static constexpr int HS_TS_WG = 8192;
    const float* unit_vec_ = nullptr;  // 1024 ones, then 1024 zeros: layer scale / bias of a fold that has none (the head-split block's output: gamma = 1)
    std::unordered_map<const void*, const void*> frag_acc_w_;  // ... -> its copy in accumulator-operand k order (Wo of the head-split block, kernels_xattn_hs.hip)
    void prepare_vocoder_constants();     // zero-latent response of the loaded model: quiet chunk and edge tail
    void prepare_ffn_weights();           // fragment-ordered copies of every ConvNeXt block's pw1 / pw2 (kernels_ffn.hip, K4)
    struct FfnW { const void* wseq = nullptr; const void* wsplit[4] = {nullptr, nullptr, nullptr, nullptr}; };  // wsplit: the hidden-split stage streams for S = 4, 12, 24, 8
    static int split_slot(int S) { return S == 4 ? 0 : S == 12 ? 1 : S == 24 ? 2 : 3; }
    std::unordered_map<const void*, FfnW> ffn_w_;  // key: the block's row-major 16-bit pw1 matrix

    // ---- host-pointer stages: 1:1 with the reference's four Run sites ------------------------------
    void duration(int B, int Lt, const int64_t* ids, const float* style_dp, const float* text_mask, float* dur);
    void text_enc(int B, int Lt, const int64_t* ids, const float* style_ttl, const float* text_mask, float* text_emb);
    void vector_est(int B, int L, int Lt, const float* noisy, const float* text_emb, const float* style_ttl,
                    const float* text_mask, const float* latent_mask, const float* total_step,
                    const float* current_step, float* denoised);
    void vocoder(int B, int L, const float* latent, float* wav);

    // ---- resident batch: upload once, run on device, fetch --------------------------------------------
    struct Batch {
        int B = 0, Lt = 0, L = 0, noise_L = 0, total_step = 0;
        float speed = 1.f;
        bool have_override = false, have_noise = false;
        uint64_t noise_seed = 0;
        // device buffers persist across uploads and only ever grow (no hipFree/hipMalloc per request; a captured graph
        // stays valid while `gen` — bumped by every reallocation — is unchanged)
        int64_t* ids = nullptr; size_t ids_cap = 0;
        int* tlen = nullptr; size_t tlen_cap = 0;
        float* style_ttl = nullptr; size_t ttl_cap = 0;
        float* style_dp = nullptr; size_t dp_cap = 0;
        float* dur = nullptr; size_t dur_cap = 0;
        int* llen = nullptr; size_t llen_cap = 0;
        int64_t* utt_ids = nullptr; size_t utt_cap = 0;
        float* noise = nullptr; size_t noise_cap = 0;   // injected noise [B,D,L] (optional)
        float* xt[2] = {nullptr, nullptr}; size_t xt_cap[2] = {0, 0};
        float* wav = nullptr; size_t wav_cap = 0;        // [B, L*cs]
        int16_t* pcm = nullptr; size_t pcm_cap = 0;      // [B, L*cs] int16 (on demand)
        // text-encoder output rows: written on the side stream (text_side), copied at the head of the main pipeline into the
        // buffer the captured graphs read (text_rows) — so the encoder of run i+1 can work while run i still reads its rows
        unsigned char* text_side = nullptr; size_t text_side_cap = 0;
        unsigned char* text_rows = nullptr; size_t text_cap = 0;
        int* toff = nullptr; size_t toff_cap = 0;       // packed text rows: first row of each utterance [B+1]
        int trows = 0;                                   // sum of the text lengths
        uint64_t gen = 0;
        std::vector<float> h_dur; std::vector<int> h_llen;
    };
    void batch_upload(int B, int Lt, const int64_t* ids, const float* text_mask, const float* style_ttl,
                      const float* style_dp, const float* duration_override, const int64_t* utt_ids);
    void batch_set_noise(const float* noise, int L);  // injected xt for the L the durations imply
    void batch_run(int total_step, float speed, uint64_t noise_seed);
    // hipGraph replay of the post-duration pipeline (text encoder, noise, Euler loop, vocoder): captured the second time
    // a shape is seen, replayed afterwards; per-call data (latent lengths, noise seed) travel through pinned host buffers.
    void set_graph_mode(bool on) { graph_on_ = on; }
    // length-aware vocoder in batch_run: every utterance's frames end at its own length, as in a batch-of-one run
    void set_vocoder_mode(bool length_aware) { vo_ragged_ = length_aware; }
    // vector-estimator row layout in batch_run: packed rows (default) or the padded [b*L + t] rows (tests compare the two)
    void set_packed_rows(bool on) { packed_ve_ = on; }
    // GELU form of the loaded model: 0 = erf (the default: torch.nn.GELU()), 1 = the tanh approximation.  stn_load_dir sets it from how the
    // graphs spell the activation (graph_bind Result::gelu); fp32 and f16 results follow it exactly, bf16 stores take the tanh-form
    // shortcut for both (|difference to erf| <= 5e-4, below half a bf16 ulp where it matters), and the fused K4 kernels compute the
    // exp2 (= tanh) form in both 16-bit modes.
    void set_gelu_form(int tanh_form) { if ((tanh_form != 0) != (gelu_act_ == ACT_GELU_TANH)) { sync(); drop_graphs(); } gelu_act_ = tanh_form ? ACT_GELU_TANH : ACT_GELU; }
    int gelu_form() const { return gelu_act_ == ACT_GELU_TANH ? 1 : 0; }
    // Shape buckets for the graph cache (the service path: requests of unlike lengths).  A captured pipeline has every size baked in:
    // B, Lt, L and the packed row counts (sum of the latent lengths, of the token counts, the vocoder's rows).  With buckets on, Lt, L and
    // the three row counts are rounded UP to a bucket boundary (bucket_up: four buckets per octave, <= 25 % padding) — the utterances'
    // own lengths stay exact and travel through device memory, so two requests that fall into the same buckets replay ONE graph.  Rows
    // behind the last sequence are dead: the row-independent kernels compute garbage there that nothing reads, the per-sequence kernels
    // never touch them, and every utterance's result over its own frames is bit-identical to the unbucketed run.  What changes for the
    // caller: stn_batch_dims reports the bucketed L (and W = L * chunk samples) and the fetched rows are that long.  B stays exact.
    // (No cached graph is dropped by the switch: a graph is fully described by the sizes in its key, bucketed or not.)
    void set_shape_buckets(bool on) { shape_buckets_ = on; }
    static int64_t bucket_up(int64_t x, int64_t min_gran) {
        if (x <= 0) return x;
        int64_t g = min_gran;
        while (g * 8 <= x) g *= 2;  // granularity = a quarter .. an eighth of x, at least min_gran: 4 - 8 buckets per octave
        return (x + g - 1) / g * g;
    }
    // measurement aid (bench.py `lone_batch_predicted_path`): with durations forced for shape control the predictor's device->host read
    // is skipped; this switch performs the read and the wait anyway, so the timed critical path is that of a predicted-duration run
    void set_duration_read(bool always) { dur_read_always_ = always; }
    // cross-attention blocks of the estimator head-split (kernels_xattn_hs.hip) instead of four launches
    void set_fused_xattn(int mode) { fused_xattn_ = mode ? 1 : 0; }  // 0: four launches; otherwise head-split (fold_ln + one launch, kernels_xattn_hs.hip)
    // K4: the pointwise pair of a ConvNeXt block as one launch.  Bit mask over the stages: 1 = vocoder, 2 = vector estimator,
    // 4 = text encoder / duration predictor.  bf16 engines, widths 256 / 384 / 512 (ffn_fused_supported)
    // 8 = the estimator's blocks as K4-split (hidden dimension cut over 4 workgroups per 128-row slab, 16-bit partial sums folded
    // by the next reader of x: fold_dwconv_ln / fold_ln); packed rows, from ffn_split_min_rows() rows on
    void set_fused_ffn(int mask) { fused_ffn_ = mask; }
    void set_fused_ffn_min_rows(int64_t k4_rows, int64_t split_rows) {
        sync(); drop_graphs();  // a captured pipeline has the kernel choice baked in
        if (k4_rows >= 0) ffn_min_rows_ = k4_rows;
        if (split_rows >= 0) ffn_split_min_rows_ = split_rows;
    }
    int fused_ffn() const { return fused_ffn_; }
    int64_t last_ve_rows() const { return last_ve_rows_; }
    int64_t last_vo_rows() const { return last_vo_rows_; }  // frames the vocoder computed in the last batch_run  // rows the estimator worked on in the last batch_run
    bool packed_text_ok(int B) const {
        return packed_ve_ && B <= 1024 && dwconv_ln_supports_packed(a_.te_dim, a_.te_kernel) && dwconv_ln_supports_packed(a_.dp_dim, a_.dp_kernel);
    }
    bool packed_rows_ok(int B) const { return packed_ve_ && B <= 1024 && a_.ve_dilated > 0 && dwconv_ln_supports_packed(a_.ve_dim, a_.ve_kernel); }
    long graph_replays() const { return graph_replays_; }
    size_t graphs_cached() const { return graphs_.size(); }
    const Batch& batch() const { return bt_; }
    void batch_fetch(float* wav, size_t wav_capacity, float* duration);
    // waveform as 16-bit PCM (clamp, *32767, truncate: cpp/helper.cpp:986-987) converted on the GPU: half the D2H bytes
    void batch_fetch_pcm16(int16_t* pcm, size_t capacity, float* duration);
    // The same, pipelined: _begin converts into the device slot and starts its device->host copy on a second stream (the next
    // stn_batch_upload / stn_batch_run proceed meanwhile: the copy of batch i overlaps the synthesis of batch i+1); _end waits
    // for that copy and hands out the slot's pinned host buffer (valid until the slot's next _begin).  Two slots.
    bool fetch_slot_dims(int slot, int* B, int64_t* W) const {
        const FetchSlot& f = fetch_[slot];
        if (!f.busy || f.dur.empty()) return false;
        *B = (int)f.dur.size(); *W = (int64_t)(f.n / f.dur.size());
        return true;
    }
    void batch_fetch_pcm16_begin(int slot);
    void batch_fetch_pcm16_end(int slot, const int16_t** pcm, size_t* n, float* duration);
    void batch_fetch_latent(float* latent);  // final denoised latent [B,D,L] (tests)
    // device->device: wav rows [B][W] into dst rows of stride dst_stride floats (>= W), on the engine's stream
    void batch_copy_wav_device(float* dst, int64_t dst_stride);
    // the finished batch as int16 PCM (writeWavFile's conversion) straight into a device buffer, rows dst_stride apart
    void batch_copy_pcm16_device(int16_t* dst, int64_t dst_stride);

    // ---- profiling (hipEvent pairs around launches of one kernel family, on this stream) ----------------
    void profile_enable(bool on) { if (on != prof_on_) profile_reset(); prof_on_ = on; }
    void launch_log_enable(bool on);   // record (family, kernel) of every launch while profiling is on (this thread's engine calls)
    std::string launch_log() const;    // "family\tkernel\n" per launch since the last profile_reset, dispatch order
    void profile_sample(int every) { prof_every_ = every < 1 ? 1 : every; prof_seen_ = 0; }  // time every n-th matching launch only
    void profile_filter(const std::string& family) { if (family != prof_filter_) profile_reset(); prof_filter_ = family; }  // "" = every family
    void profile_reset();
    std::vector<std::pair<std::string, KernelStat>> profile_collect();

    // ---- op-level test entry points (host pointers) ------------------------------------------------------
    void op_gemm(int dtype, int M, int N, int K, const float* A, const float* W, const float* bias, int act, float* out);
    void op_attention(int dtype, int B, int Lq, int Lk, int H, int dh, const float* q, const float* k, const float* v,
                      const int* qlen, const int* klen, int rope_mode, float* o);
    void op_dwconv_ln(int dtype, int B, int L, int C, int k, int dil, const float* x, const float* w, const float* bias,
                      const float* g, const float* b, float* y, const int* seqlen = nullptr /* host [B] */);
    // device-resident timing of one GEMM shape (random operands), HIP events around `iters` launches: avg ms
    double op_gemm_bench(int dtype, int M, int N, int K, int mode, int iters);
    // per-workgroup phase stamps of one launch of the tiled kernel: out[0..2] = mean cycles of (first stage landed, K loop,
    // epilogue), out[3] = max over workgroups of (end - earliest entry), out[4] = spread of entry times, out[5] = workgroups
    void op_gemm_phases(int dtype, int M, int N, int K, int mode, double* out6);
    void op_randn(uint64_t seed, int B, int D, int L, const int64_t* utt_ids, const int* len, float* out);
    // K4 on host operands: x <- x + gamma * (W2 . GELU(W1 . xn + b1) + b2) [+ rowvec[row_b[m]]]; xn is rounded to the engine's
    // 16-bit format first.  fused = false runs the two tiled GEMM launches on the same operands (the pair K4 replaces).
    // mode: 0 = two tiled launches, 1 = K4, 2 = K4-split (partial sums) + fold_ln
    void op_ffn(int M, int C, int I, const float* xn, const float* W1, const float* b1, const float* W2, const float* b2, const float* gamma,
                const float* rowvec /* [nseq][C] or null */, const int* row_b /* host [M] or null */, int nseq, float* x, int mode);
    // fold_dwconv_ln on host operands (packed rows: sequence b owns seqlen[b] consecutive rows; M = sum): part [S][M][C] fp32 is rounded
    // to the engine's 16-bit format first.  x_out <- folded x, y <- LayerNorm(dwconv(folded x)) as fp32.
    void op_fold_dwconv_ln(int B, int C, int k, int dil, int S, const int* seqlen, const float* x, const float* part, const float* b2,
                           const float* gamma, const float* rowvec /* [B][C] or null */, const float* w /*[C][k]*/, const float* bias,
                           const float* g, const float* b, float* x_out, float* y);
    // device-resident timing of the block's pointwise pair on random operands: out[0] = avg ms per call (fused: one launch,
    // unfused: pw1 + pw2); fused only: out[1..3] = mean cycles per workgroup to the first stage / in the tile loop / in the
    // epilogue, out[4] = workgroups
    void op_ffn_bench(int M, int C, int I, int mode, int iters, double* out5);
    // device-resident timing of one estimator-style block chain on packed rows of B equal sequences of L frames:
    // mode 0: dwconv_ln + pw1 + pw2;  mode 2: fold_dwconv_ln + K4-split.  out[0] = avg ms per block, out[1] = avg ms of the conv kernel alone, out[2..5] = fold_dwconv_ln phase cycles (mode 2)
    void op_block_bench(int B, int L, int C, int I, int k, int dil, int mode, int iters, double* out6);

    Arena& arena() { return ar_; }

   private:
    // weights
    DevTensor& tensor(const std::string& name);
    const float* vecf(const std::string& name) { return tensor(name).f32; }
    Linear linear(const std::string& prefix);
    LNorm lnorm(const std::string& prefix);
    ConvNeXt convnext_w(const std::string& prefix);
    Attn attn_w(const std::string& prefix, bool self);
    void free_weights();
    using RawSource = std::function<std::vector<float>(const std::string& name, int kind, int rows, int cols, float gain)>;
    void load_weights(const stn_arch& a, const RawSource& src, std::vector<std::string>* names_only);

    // building blocks (enqueue on s_)
    size_t act_bytes(int64_t n) const { return (size_t)n * (is_half(dt_) ? 2 : 4); }
    void* act_alloc(int64_t n) { return ar_.alloc(act_bytes(n)); }
    float* f32_alloc(int64_t n) { return static_cast<float*>(ar_.alloc((size_t)n * 4)); }
    void gemm(const char* tag, int dt, const void* A, int lda, const Linear& w, int M, Epilogue e);
    // rowvec (optional, [B][rv_ld]): added to every row of sequence b in the same residual epilogue (time conditioning)
    // The residual stream of a stage whose blocks may run as K4-split: x plus at most one pending (unfolded) update.
    struct FoldState {
        float* x = nullptr;      // current residual rows
        float* x_alt = nullptr;  // second buffer: fold_dwconv_ln writes the folded stream there and the two swap
        void* part = nullptr;    // [S][rows padded to 128][C] 16-bit partial sums (one buffer: its reader runs before the next writer)
        int64_t part_stride = 0;
        int S = 0;               // the split this stage's launches take (ffn_split_choose of its row count)
        bool pending = false;
        FoldArgs fold;           // the pending update
    };
    // fs (optional): x is fs->x; the block may leave its pointwise pair pending in fs (the caller folds it with the next reader of x)
    void convnext(const ConvNeXt& p, float* x, int B, int L, int C, int hid, int k, int dil, const int* len,
                  const int* conv_len = nullptr, const float* rowvec = nullptr, int rv_ld = 0, const Ragged* rg = nullptr, FoldState* fs = nullptr);
    // LayerNorm of the stage's residual stream into xn, folding a pending update first
    void fold_layernorm(FoldState& fs, int64_t M, int C, const LNorm& ln, void* xn, const char* tag);
    void attn_block(const Attn& p, float* x, int B, int Lq, int C, int H, const void* ctx, int Lk, const int* qlen,
                    const int* klen, int rope_mode, bool self, const Ragged* qrg = nullptr);
    void* to_act(const float* src, int64_t n);

    struct ProfSpan { std::string tag; hipEvent_t a, b; double flops, bytes; };
    void prof_begin(const char* tag, double flops, double bytes);
    void prof_end();

    int device_, dt_;
    int n_cu_ = 256;  // compute units of the device (launch shapes that depend on rounds of workgroups)
    hipStream_t s_ = nullptr, own_s_ = nullptr;
    const char* stage_ = "";
    std::string prof_filter_;
    std::string log_family_;
    int prof_every_ = 1;
    uint64_t prof_seen_ = 0;
    bool prof_active_ = false;
    stn_arch a_{};
    bool loaded_ = false;
    int64_t params_ = 0;
    std::unordered_map<std::string, DevTensor> w_;
    std::vector<void*> owned_;
    Arena ar_;
    Batch bt_;
    std::vector<void*> batch_owned_;    // every live batch buffer (freed with the engine)
    std::vector<void*> batch_retired_;  // outgrown buffers: freed at the next upload, after the stream has drained
    template <typename T> void ensure(T*& p, size_t& cap, size_t need);
    std::vector<float> reported_dur_;
    void enqueue_after_duration(int total_step, const std::function<void()>& take_text_rows);
    // A captured graph holds raw pointers: everything it can have baked in is part of its key — the batch buffers (`gen`, bumped
    // by every reallocation), the weights (`wgen`, bumped by every load), the pinned staging the copy nodes read, the stream.
    struct GraphKey {
        int B = 0, Lt = 0, L = 0, steps = 0; bool noise = false, ragged = false; int xattn = 0; int ffn = 0; int rows = 0, vrows = 0, trows = 0;
        uint64_t gen = 0, wgen = 0; const void* p0 = nullptr; const void* p1 = nullptr; const void* pin = nullptr; hipStream_t s = nullptr;
        bool operator==(const GraphKey& o) const {
            return B == o.B && Lt == o.Lt && L == o.L && steps == o.steps && noise == o.noise && ragged == o.ragged && xattn == o.xattn && ffn == o.ffn &&
                   rows == o.rows && vrows == o.vrows && trows == o.trows && gen == o.gen && wgen == o.wgen && p0 == o.p0 && p1 == o.p1 && pin == o.pin && s == o.s;
        }
    };
    // LRU cache of captured graphs (the reference's call() loop and the n_test loop alternate a few shapes,
    // /root/reference/cpp/helper.cpp:697-719, cpp/example_onnx.cpp:88): a shape is captured the second time it is seen
    // (`warm_keys_`: the arena must have seen its allocation sequence once) and replayed from then on.
    // (graph / exec: the pipeline up to the first text cross-attention; graph2 / exec2: the rest — the text encoder's rows are taken between the two)
    struct GraphEntry { GraphKey key; hipGraph_t graph = nullptr, graph2 = nullptr; hipGraphExec_t exec = nullptr, exec2 = nullptr; uint64_t last_use = 0; };
    static constexpr size_t kGraphCache = 8, kWarmKeys = 16;
    std::vector<GraphEntry> graphs_;
    std::vector<GraphKey> warm_keys_;
    uint64_t graph_clock_ = 0, wgen_ = 0;
    void drop_graphs();  // destroy every cached graph (weights or staging they point at are going away)
    bool graph_on_ = true;
    bool vo_ragged_ = false;
    bool packed_ve_ = true;
    bool nt_hints_ = true;
    // The duration predictor of stn_batch_run depends on the uploaded inputs only (not on the previous batch, not on the text
    // encoder): it runs on its own stream with its own workspace, beside the text encoder of the same batch and, in back-to-back
    // runs, beside the tail of the previous one.  Everything that overwrites its inputs goes through sync(), which waits for both.
    hipStream_t dp_s_ = nullptr, te_s_ = nullptr;
    Arena dp_ar_, te_ar_;
    // The text encoder depends on the uploaded inputs only as well: it runs on the same side stream (after the predictor) into
    // text_side; the main stream waits for ev_te_, copies the rows into text_rows (what the graphs read) and records ev_copied_,
    // which the side stream waits for before the NEXT run's encoder overwrites text_side (prefetch depth: one run).
    hipEvent_t ev_te_ = nullptr, ev_copied_ = nullptr, ev_dp_ = nullptr;
    std::function<void()> text_gate_;  // armed by enqueue_after_duration, fired once by the first text cross-attention of the run
    bool copied_valid_ = false;
    int64_t ffn_gate_rows_ = 0;     // row count the K4 decision is taken on when it is not the launch's own (trimmed dense vocoder)
    int64_t ffn_min_rows_ = 18432;  // K4 only from this many rows on (144 workgroups); STN_FFN_MIN_ROWS overrides
    int64_t ffn_split_min_rows_ = 1;     // K4-split from this many rows on (with the slab staged through LDS one utterance gains too: 20.0 vs 21.4 us per block); STN_FFN_SPLIT_MIN_ROWS overrides
    int fused_ffn_ = 9;         // K4 stages (set_fused_ffn): adopted where measured faster; STN_FFN=<mask> overrides
    bool shape_buckets_ = false;
    bool dur_read_always_ = false;  // measurement: read the predicted durations back (and wait for them) even when the caller overrides them
    int gelu_act_ = ACT_GELU;
    int fused_xattn_ = 1;  // cross-attention blocks of the estimator: head-split (fold_ln + one launch, kernels_xattn_hs.hip; the default) or, 0, four
                           // launches; STN_XATTN=<0|1> overrides
    int64_t last_ve_rows_ = 0, last_vo_rows_ = 0;
    float* vo_quiet_ = nullptr;  // [base_chunk_size]      } zero-latent response of the vocoder (device, owned), 16-bit engines
    float* vo_edge_ = nullptr;   // [rf][base_chunk_size]  }
    int vo_rf_ = 0;
    int trimmed_rows(int B, int L, std::vector<int>* n_host) const;  // sum of the trimmed extents (0: trimming not applicable)
    long graph_replays_ = 0;
    int* pin_llen_ = nullptr; size_t pin_llen_cap_ = 0;   // pinned host staging read by the graph's memcpy node
    unsigned long long* pin_seed_ = nullptr;
    bool pin_valid_ = false;
    unsigned long long* seed_dev_ = nullptr;
    int final_xt_ = 0;
    struct FetchSlot { int16_t* dev = nullptr; int16_t* pin = nullptr; size_t cap = 0, n = 0; hipEvent_t ready = nullptr, done = nullptr; bool busy = false; std::vector<float> dur; };
    FetchSlot fetch_[2];
    hipStream_t copy_s_ = nullptr;
    bool prof_on_ = false;
    std::vector<ProfSpan> spans_;
    std::vector<hipEvent_t> ev_pool_;
};

}  // namespace stn
