// kernels.hpp — launch interface of the hand-written gfx950 kernels (kernels_*.hip).
// Activation layout everywhere: row-major [rows = b*len + t][channels].
// The residual stream is always fp32; "operand" activations (LayerNorm outputs, GELU hidden,
// q/k/v/o) are `act_t` = float (STN_F32) or bf16 (STN_BF16).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <stdexcept>
#include <utility>
#include <vector>
#include <string>

namespace stn {

// Kernel-exact timing: when a pair of events is armed here, the NEXT kernel launched through STN_KLAUNCH carries them on
// its own dispatch packet (hipExtLaunchKernelGGL: start/stop are the kernel's begin/end timestamps, what rocprofv3 reports),
// instead of being bracketed by hipEventRecord barriers that add the dispatch boundary to the span.
struct LaunchEvents { hipEvent_t start = nullptr, stop = nullptr; };
inline thread_local LaunchEvents g_launch_ev;
// Launch log (profiling runs only): kernel expression + the kernel family the engine was in, one entry per launch in dispatch
// order — what tools/pmc_families.py aligns rocprofv3's per-dispatch rows with, so that counters are attributed to families
// by position instead of by guessing from template arguments and grid sizes.
struct LaunchLog {
    bool on = false;
    const char* family = "-";
    std::vector<std::pair<std::string, const char*>> entries;  // (family, kernel expression)
    void add(const char* kexpr) { if (entries.size() < 200000) entries.emplace_back(family, kexpr); }
};
inline thread_local LaunchLog g_launch_log;
#define STN_KLAUNCH(kernel, grid, block, lds, stream, ...)                                                              \
    do {                                                                                                                \
        if (::stn::g_launch_log.on) ::stn::g_launch_log.add(#kernel);                                                   \
        if (::stn::g_launch_ev.start) {                                                                                 \
            const ::stn::LaunchEvents ev_ = ::stn::g_launch_ev;                                                         \
            ::stn::g_launch_ev = ::stn::LaunchEvents{};                                                                 \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, ev_.start, ev_.stop, 0, __VA_ARGS__);               \
        } else {                                                                                                        \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                          \
        }                                                                                                               \
    } while (0)

// hipFuncSetAttribute is per device: one flag per (kernel instantiation, device) so that several handles on different GPUs in
// one process each opt their device in (flags are only ever set to true, so a race merely repeats the call)
struct PerDeviceOnce {
    bool done[64] = {};
    bool need() { int d = 0; (void)hipGetDevice(&d); d &= 63; if (done[d]) return false; done[d] = true; return true; }
};

// launch-side HIP calls that must not fail silently (a failed attribute call leaves a sticky error that surfaces in
// whatever library checks hipGetLastError next).  Nothing in this library calls abort(): contract violations and HIP failures
// are exceptions, which the C ABI (api.cpp) turns into STN_ERR_INVALID / STN_ERR_DEVICE + stn_last_error — the reference's
// hosts get std::runtime_error from the same situations (/root/reference/cpp/helper.cpp:479-481, 193).
inline void stn_check_hip(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string("HIP error: ") + what + " failed: " + hipGetErrorString(e));
}


enum DType : int { F32 = 0, BF16 = 1, F16 = 2 };  // F16: IEEE half operands / activations (v_mfma_f32_32x32x16_f16), fp32 accumulate
__host__ __device__ inline bool is_half(int dt) { return dt == BF16 || dt == F16; }  // 2-byte storage
typedef _Float16 f16_t;                                          // storage type of the F16 mode (bf16 is raw uint16_t)
enum ActFn : int { ACT_NONE = 0, ACT_GELU = 1, ACT_SILU = 2, ACT_GELU_TANH = 3 };  // GELU: erf form; GELU_TANH: 0.5 x (1 + tanh(sqrt(2/pi)(x + 0.044715 x^3)))

// GEMM epilogue description:  acc[m][n] = sum_k A[m][k] * W[n][k]
enum EpiMode : int {
    EPI_STORE = 0,    // out[m*ldo + n] = act(acc + bias[n]) * rowmask(m)        (out dtype = out_dtype)
    EPI_RESID = 1,    // resid[m*ldo+n] = (resid + gamma[n]*(acc+bias[n])) * rowmask(m)      (fp32, in place)
    EPI_EULER_T = 2,  // out[b][n][t] = (aux[b][n][t] + (acc+bias[n]) * row_scale[b]) * rowmask   (fp32, [B,N,L])
    EPI_STORE_T = 3,  // out[b][n][t] = (acc+bias[n]) * rowmask                                  (fp32, [B,N,L])
};

struct Epilogue {
    int mode = EPI_STORE;
    int act = ACT_NONE;
    int out_dtype = F32;         // EPI_STORE only
    const float* bias = nullptr; // [N] or null
    void* out = nullptr;         // EPI_STORE / *_T destination
    int ldo = 0;                 // row stride of out / resid (elements)
    const float* gamma = nullptr;// [N] layer-scale (EPI_RESID) or null
    float* resid = nullptr;      // EPI_RESID
    const int* len = nullptr;    // [B] valid rows per sequence (null -> no row mask)
    int L = 1;                   // rows per sequence (row m -> b = m / L, t = m % L)
    const float* aux = nullptr;  // EPI_EULER_T: previous latent [B,N,L]
    const float* row_scale = nullptr; // EPI_EULER_T: per-b scale (dt)
    const int* row_b = nullptr;       // packed rows: sequence of row m (replaces m / L for rowvec; len must be null then)
    const float* rowvec = nullptr;    // EPI_RESID: per-sequence vector [B][rv_ld] added to every row of sequence b (time
    int rv_ld = 0;                    //            conditioning): resid = (resid + gamma*(acc+bias) + rowvec[b]) * keep
    int ksplit = 1;                   // tiled kernels: split-K factor (plain fp32 store of partials; launch_gemm_splitk)
    int nt = 0;                       // tiled kernels: 1 = non-temporal 16-bit output stores, for a stream far larger than the Infinity Cache
                                      // (the vocoder's 245 MB hidden activation): vo.pw1 185 -> 166 us.  The matching non-temporal A loads in
                                      // pw2 were measured too and cost +9 % (each A panel is read by two column tiles), non-temporal loads of the old residual in pw2's
                                      // epilogue +3 %: so there are none
    int tr_epilogue = 0;              // tiled kernels, bf16 store: wave-private transposed-image epilogue (set by the launcher)
    unsigned long long* ts = nullptr; // diagnostics (tiled kernels): 4 shader-clock stamps per workgroup — entry, first
                                      // stage landed, K-loop done, epilogue done (stn_op_gemm_phases)
};

// A: [M][lda] (dtype), W: [N][ldw] (same dtype), K % 8 == 0 (bf16) / K % 4 == 0 (f32), 16-byte aligned rows.
void launch_gemm(hipStream_t s, int dtype, const void* A, int lda, const void* W, int ldw, int M, int N, int K,
                 const Epilogue& e);
// Deterministic split-K for tiny-M, long-K GEMMs: gemm_splitk_factor says how many ways to split (1 = don't);
// launch_gemm_splitk runs the splits into `workspace` ([S][M][N] fp32) and reduces them in split order with epilogue e.
int gemm_splitk_factor(int dtype, int M, int N, int K, const Epilogue& e);
void launch_gemm_splitk(hipStream_t s, int dtype, const void* A, int lda, const void* W, int ldw, int M, int N, int K, const Epilogue& e,
                        int S, float* workspace);

// K4 — the pointwise pair of a ConvNeXt block in one launch (kernels_ffn.hip):
//   x[m][:] <- (x[m][:] + gamma * (W2 . GELU(W1 . xn[m] + b1) + b2) + rowvec[seq(m)]) * keep(m)
// 16-bit modes; W1 / W2 pre-packed once at model load (launch_repack_frag / launch_repack_frag_acc).
struct FfnArgs {
    const void* xn = nullptr; int ldx = 0;        // LayerNorm output [M][ldx], 16-bit activation format
    const void* wseq = nullptr;                   // both matrices in fragment order, interleaved in the order the kernel consumes
                                                  // them (launch_ffn_pack): 2 * I * C 16-bit values
    const float* b1 = nullptr;                    // [I]
    const float* b2 = nullptr;                    // [C] or null
    const float* gamma = nullptr;                 // [C] layer scale or null
    float* x = nullptr; int ldo = 0;              // residual stream [M][ldo] fp32, updated in place
    int M = 0, I = 0;
    const int* row_b = nullptr;                   // packed rows: sequence of row m (for rowvec)
    const float* rowvec = nullptr; int rv_ld = 0; // per-sequence vector added to every row (time conditioning) or null
    const int* len = nullptr; int L = 1;          // padded rows: row m = b*L + t is zeroed when t >= len[b] (null: no mask)
    unsigned long long* ts = nullptr;             // diagnostics: 4 shader-clock stamps per workgroup (entry, first stage, loop, end)
    // hidden split (ffn_split_factor(C, I) = S > 1): S workgroups per 128-row slab, each over I/S hidden units; wseq packed with
    // the same S (launch_ffn_pack); the result is NOT applied to x: part[sp][m][:] (16-bit, [S][part_stride / C rows][C]) receives
    // W2[:, hidden share sp] . GELU(W1[hidden share sp] . xn[m] + b1) and launch_fold_ln / launch_fold_dwconv_ln fold
    // x <- x + gamma * (sum_sp part[sp] + b2) + rowvec into the residual stream.  b2 / gamma / x / rowvec / len are unused here.
    int split = 1;
    void* part = nullptr;                         // rows padded to a multiple of 128
    int64_t part_stride = 0;                      // elements between two splits (= padded rows * C)
};
bool ffn_fused_supported(int dtype, int C, int I);
// whether the block shape has a K4-split form at all (0: no; > 1: yes)
int ffn_split_factor(int dtype, int C, int I);
// splits a block shape can run with (4, 8, 12, 24: I / 32 / S hidden tiles per workgroup, an even number), and the one a launch of M rows
// takes: 12 ways up to 12 slabs (1536 rows), 8 ways up to 32 slabs (4096 rows: still one round of workgroups), 4 ways beyond.  The split is a function of the launch's ROW COUNT, so the order in which a
// row's 16-bit partial sums are added depends on how many rows the launch has: results on either side of the boundary (an utterance
// alone, a 16-utterance shard of a strong-scaling run, the unsharded batch) agree to rounding, not bit for bit (include/stn.h,
// tests/test_gpu_ffn.py::test_ffn_split_regimes_agree_to_rounding: 1 536, 1 664 and 4 224 rows)
bool ffn_split_valid(int dtype, int C, int I, int S);
int ffn_split_choose(int dtype, int C, int I, int64_t M);
inline int64_t ffn_split_rows(int64_t M) { return (M + 127) / 128 * 128; }
// K4's LDS layout, one definition for the kernel and its launcher: 4 ring buffers of C * 64 bytes | b2, gamma (C floats each) | b1 (I floats) |
// at C = 384 a fifth buffer, KiB-aligned, for wave 0's rows of the slab in the prologue (kernels_ffn_body.inc)
__host__ __device__ constexpr inline int ffn_lds_side_offset(int C, int I) { return (4 * C * 64 + (I + 2 * C) * 4 + 1023) & ~1023; }
__host__ __device__ constexpr inline bool ffn_lds_has_side(int C) { return C == 384; }
void launch_ffn_fused(hipStream_t s, int dtype, int C, const FfnArgs& a);
// The pending update of a K4-split launch, folded by the next reader of x:
//   x_new[m] = x[m] + gamma * (((part[0][m] + part[1][m]) + part[2][m]) + ... + b2) + rowvec[seq(m)]      (fp32, this order)
struct FoldArgs {
    const void* part = nullptr; int S = 0; int64_t part_stride = 0;   // 16-bit partial sums (the engine's activation format)
    const float* b2 = nullptr;        // [C] or null
    const float* gamma = nullptr;     // [C] or null (1)
    const float* rowvec = nullptr; int rv_ld = 0;  // per-sequence vector or null
    const int* row_b = nullptr;       // fold_ln with rowvec: sequence of row m
    unsigned long long* ts = nullptr; // diagnostics (fold_dwconv_ln): 4 shader-clock stamps per workgroup — entry, phase 1 done, hand-over, end
    int run_frames = 0;               // fold_dwconv_ln: longest run of frames a workgroup takes (0: the default, 32; fold_run_frames() picks it from the lengths)
};
// fold_dwconv_ln puts ONE 1024-thread workgroup on a CU, so a launch of 257 workgroups takes two rounds where 256 take one (12.6 -> 16.2 us at
// the bench's shape).  The run length whose GRID (B x ceil(longest / run): the placeholders of runs a sequence does not have take a CU each on their way
// out) fits ONE round of `n_cu` workgroups — 40 or 48 frames where 32 does not; 0 (the default) otherwise: with several rounds either way the count
// of rounds stops predicting the time.  A frame's arithmetic does not depend on the run it falls in: same bits for any choice.
int fold_run_frames(const int* lengths, int B, int n_cu);
// x (in place) <- folded x;  y <- LayerNorm(folded x)       (rows are independent)
void launch_fold_ln(hipStream_t s, int act_dtype, float* x, int64_t M, int C, const FoldArgs& f, const float* g, const float* b, float eps, void* y);
// packed rows only (row_off / seqlen as launch_dwconv_ln): x_out <- folded x_in (x_out != x_in: a workgroup re-folds the halo rows
// of its neighbours);  y <- LayerNorm(dwconv(folded x)).  k = 5 or 7, C % 8 == 0, C <= 512.
bool fold_dwconv_ln_supported(int C, int k, int dil);
void launch_fold_dwconv_ln(hipStream_t s, int act_dtype, const float* x_in, float* x_out, int B, int L, int C, const FoldArgs& f, const float* w_t,
                           const float* bias, int k, int dil, const float* ln_g, const float* ln_b, float eps, void* y, const int* seqlen,
                           const int* row_off);
// W [N][K] row-major 16-bit -> A fragments whose k order matches a GELU'd accumulator used as the B operand (N % 32, K % 32 == 0)
void launch_repack_frag_acc(hipStream_t s, const void* W, int N, int K, void* Wf);
// W1 [I][C], W2 [C][I] (row-major 16-bit) -> wseq: hidden tile t of W1 as C/16 KiB fragment pieces, hidden tile t of W2 as
// C/16 KiB pieces in the accumulator-operand order, laid out as the stage sequence W1(0), W1(1), W2(0), W1(2), W2(1), ...,
// W1(T-1), W2(T-2), W2(T-1) (T = I/32): the kernel's LDS-DMA stream is then one linear walk through memory.
// tmp: 2 * I * C 16-bit values of scratch.
// S > 1 (hidden split): S such streams one after another, stream sp over hidden tiles [sp*T/S, (sp+1)*T/S).
void launch_ffn_pack(hipStream_t s, const void* W1, const void* W2, int C, int I, void* tmp, void* wseq, int S = 1);

// depthwise 'same' conv (taps k, dilation dil, weights TRANSPOSED [k][C]) fused with LayerNorm over C.
// x fp32 [B*L][C] -> y act [B*L][C].  C % 4 == 0, C <= 1024.
// seqlen (optional, [B]): frames at t >= seqlen[b] are treated as outside the sequence (zero taps, rows not written) — the
// length-aware mode in which a padded batch reproduces what each sequence would give on its own
void launch_dwconv_ln(hipStream_t s, int out_dtype, const float* x, int B, int L, int C, const float* w_t,
                      const float* bias, int k, int dil, const float* ln_g, const float* ln_b, float eps, void* y,
                      const int* seqlen = nullptr, const int* row_off = nullptr);
// Packed ("ragged") rows: sequence b owns rows row_off[b] .. row_off[b] + len[b] of x / y and nothing else — no padding rows
// exist.  row_off has B+1 entries (launch_row_map); supported where dwconv_ln_supports_packed(C, k).
bool dwconv_ln_supports_packed(int C, int k);
void launch_row_map(hipStream_t s, const int* len, int B, int* row_off /*[B+1]*/, int* row_b /*[sum len] or null*/,
                    int rows_padded = 0 /* row_b has this many entries: those behind sum len are set to sequence 0 (shape buckets) */);
// plain LayerNorm over C: x fp32 -> y act
void launch_layernorm(hipStream_t s, int out_dtype, const float* x, int64_t M, int C, const float* g, const float* b,
                      float eps, void* y);

// fused attention core: softmax(rope(q) rope(k)^T / sqrt(dh)) v, per (b, head).
// q [B*Lq][ldq], k/v [B*Lk][ldk] (act dtype); o [B*Lq][ldo] (act dtype).
// rope_mode: -1 none, 0 position index, 1 length-aware (gamma * t / len).
// k_rotated: the keys already carry their rotation (launch_rope_rows ran on them once) — only q is rotated here.
void launch_attention(hipStream_t s, int dtype, const void* q, int ldq, const void* k, const void* v, int ldk, void* o,
                      int ldo, int B, int Lq, int Lk, int H, int dh, const int* qlen, const int* klen, int rope_mode,
                      float rope_base, float rope_gamma, bool k_rotated = false,
                      const int* q_off = nullptr /* packed query/output rows: sequence b starts at q_off[b], owns qlen[b] */,
                      const int* k_off = nullptr /* packed key/value rows: sequence b starts at k_off[b], owns klen[b] */);
// [N][K] row-major 16-bit -> MFMA fragment order (one contiguous KiB per operand; kernels_ffn.hip), once at model load
void launch_repack_frag(hipStream_t s, const void* W, int N, int K, void* Wf);
// One launch per cross-attention block of the vector estimator, HEAD-SPLIT (kernels_xattn_hs.hip): one launch computes, per head h, q_h = Wq_h . xn + bq_h, its rotation, the attention
// over the given K / V and the head's share of the output projection, and stores it as a 16-bit partial sum part[h][row][:] in K4-split's
// layout; x <- x + ((p0 + p1) + p2) + p3 + bo is applied by the next reader of x through FoldArgs{part, S = 4, b2 = bo} (launch_fold_dwconv_ln
// / launch_fold_ln).  xn = LayerNorm(x) rows (16-bit, [M][C]); WqF = launch_repack_frag(Wq), WoA = launch_repack_frag_acc(Wo), both [C][C];
// packed query rows (q_off, qlen) only; keys already rotated when rope_mode >= 0.  ts: optional 8 shader-clock stamps per workgroup.
bool xattn_hs_supported(int dtype, int C, int H, int L, int Lk, int ldk);
void launch_xattn_hs(hipStream_t s, int dtype, const void* xn, int64_t M, const void* WqF, const float* bq, const void* kp, const void* vp, int ldk,
                     const void* WoA, void* part, int64_t part_stride, int B, int L, int Lk, const int* qlen, const int* klen,
                     const int* q_off, const int* k_off, int rope_mode, float rope_base, float rope_gamma, unsigned long long* ts = nullptr,
                     const int* pairs = nullptr /* launch_xattn_hs_pairs' table: which two utterances share a workgroup (null: 2g, 2g + 1) */);
// utterances per workgroup a launch of this shape takes (1 or 2), and the pairing for 2: sorted by length, longest with shortest, so that
// the pairs' row-tile counts are as equal as the batch allows (pairs: 2 * ceil(B / 2) ints; once per synthesis, the lengths do not change)
int xattn_hs_group(int B, int L, int Lk);
void launch_xattn_hs_pairs(hipStream_t s, const int* qlen, int B, int* pairs);
// in-place RoPE of `groups` key blocks per row: element (row b*L+t, column g*group_stride + h*dh + i) for t < len[b]
// (len null: all rows).  Keys that are reused by many attention launches (the vector estimator's text keys: every
// block of every Euler step) are rotated once here instead of at every launch.  Same arithmetic as the attention
// kernels' staging (fp32 rotation, one rounding to the storage dtype).
void launch_rope_rows(hipStream_t s, int dtype, void* x, int ld, int B, int L, const int* len, int groups, int group_stride,
                      int H, int dh, int rope_mode, float rope_base, float rope_gamma, const int* row_off = nullptr /* packed rows */);

// embedding gather: x[b*L+t][:] = (t < len[b] && 0 <= id < vocab) ? emb[id][:] : 0     (fp32 out)
void launch_embed(hipStream_t s, const int64_t* ids, const float* emb, int vocab, int B, int L, int C, const int* len,
                  float* x, const int* row_off = nullptr /* packed destination rows */);
// prefix mask [B][L] (float) -> len[B] (count of entries > 0.5)
void launch_mask_to_len(hipStream_t s, const float* mask, int B, int L, int* len);
// [B][C][L] fp32 -> rows [B*L][ld_out] act (columns C..ld_out-1 are zero-filled: K padding for the GEMM that follows)
void launch_ncl_to_rows(hipStream_t s, int out_dtype, const float* in, int B, int C, int L, void* out, int ld_out = 0,
                        const int* len = nullptr, const int* row_off = nullptr /* packed destination rows */);
// Euler update with the [B*L][D] -> [B][D][L] transpose: out[b][d][t] = t < len[b] ? prev[b][d][t] + v[(b*L+t)*D + d] * dt[b] : 0
// z_rows (optional): the new latent also as rows [row][ldz] in `z_dtype` (columns < D only; the K padding beyond stays as the caller zeroed it) — what
// launch_ncl_to_rows(out) would write, so that the next step's input projection needs no such launch
void launch_euler_ncl(hipStream_t s, const float* prev, const float* v, const float* dt, const int* len, int B, int D, int L, float* out,
                      const int* row_off = nullptr /* v in packed rows */, void* z_rows = nullptr, int z_dtype = 0, int ldz = 0);
// fp32 -> act dtype copy (n elements)
void launch_cast(hipStream_t s, int out_dtype, const float* in, int64_t n, void* out);
// x[b*L+t][c] += v[b*ldv + c] for t < len[b]
void launch_add_rowvec(hipStream_t s, float* x, const float* v, int ldv, int B, int L, int C, const int* len);
// time embedding: te[b][:] = [sin(t*f_i), cos(t*f_i)], t = cur[b]/tot[b]*scale
void launch_time_embed(hipStream_t s, const float* cur, const float* tot, int B, int dim, float scale, float* te);
// vocoder front: un-compress [B,D,L] -> frames [B*T][ld] and conv1d ld->C (kernel k, 'same'), fp32 out
void launch_scale_len(hipStream_t s, const int* len, int B, int factor, int* out);  // out[b] = len[b] * factor
void launch_vocoder_in(hipStream_t s, const float* latent, int B, int L, int ld, int ccf, const float* w /*[C][ld][k]*/,
                       const float* bias, int C, int k, float* x, const int* seqlen = nullptr);
// vocoder front as im2col for the MFMA path: cols[r][ci*k + j] = frame(t + j - k/2)[ci] (0 outside the sequence / beyond ld*k),
// frame (b, t = l*ccf + q) channel c <- latent[b][q*ld + c][l]; row stride kp (>= ld*k, zero padded), act dtype
void launch_vocoder_im2col(hipStream_t s, int out_dtype, const float* latent, int B, int L, int ld, int ccf, int k, int kp, void* cols,
                           const int* seqlen = nullptr /* valid vocoder frames per sequence */,
                           const int* row_off = nullptr /* packed destination rows (needs seqlen) */);
// packed rows [sum len][W] -> padded [B][T][W], zeros past each sequence's length
void launch_unpack_rows(hipStream_t s, const float* src, const int* len, const int* row_off, int B, int T, int W, float* dst);
// exact trimmed dense vocoder (see kernels_misc.hip): extents per utterance, and the unpack that fills the position-independent tail
void launch_trim_len(hipStream_t s, const int* len, int B, int ccf, int T, int rf, int* n_out, int* valid_out);
void launch_unpack_rows_quiet(hipStream_t s, const float* src, const int* valid, const int* row_off, int B, int T, int W, int rf,
                              const float* quiet, const float* edge, float* dst);
// masked mean over valid rows: pooled[b][c] = sum_{t<len[b]} x[b*L+t][c] / max(len[b],1)   (x act dtype, out fp32)
void launch_masked_mean(hipStream_t s, int in_dtype, const void* x, int B, int L, int C, const int* len, float* pooled,
                        const int* row_off = nullptr /* packed source rows */);
// y = softplus(x) elementwise (n small)
void launch_softplus(hipStream_t s, float* x, int n);
// durations: d[b] = (override ? override[b] : d[b]) / speed ; computes per-b latent length; see engine
void launch_scale(hipStream_t s, float* x, int n, float mul);
// Philox4x32-10 + Box-Muller noise, masked: xt[b][d][t] = t < len[b] ? N(0,1) : 0
void launch_randn_masked(hipStream_t s, uint64_t seed, const int64_t* utt_ids, int B, int D, int L, const int* len,
                         float* xt, const unsigned long long* seed_dev = nullptr /* read the seed from device memory (graph replay) */);
// xt[b][d][t] *= (t < len[b])
void launch_mask_ncl(hipStream_t s, float* x, int B, int D, int L, const int* len);
// out[i] = 1 / in[i]
void launch_reciprocal(hipStream_t s, const float* in, int n, float* out);
// bf16 / f16 -> fp32 copy (tests)
void launch_bf16_to_f32(hipStream_t s, const uint16_t* in, int64_t n, float* out);
void launch_half_to_f32(hipStream_t s, int in_dtype, const void* in, int64_t n, float* out);
// total_step/current_step helper: fill n floats
void launch_fill(hipStream_t s, float* x, int n, float v);
void launch_step_counters(hipStream_t s, float* tot /*[steps][B]*/, float* cur /*[steps][B]*/, float* dt /*[B]*/, int B, int steps);
// waveform epilogue: pcm[i] = int16(clamp(w[i], -1, 1) * 32767)  (truncation, cpp/helper.cpp:986-987)
// rows x W samples -> int16 PCM rows at pcm + row * dst_stride (dst_stride >= W)
void launch_f32_to_pcm16(hipStream_t s, const float* w, int64_t rows, int W, int16_t* pcm, int64_t dst_stride);

}  // namespace stn
