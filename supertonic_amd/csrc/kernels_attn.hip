// kernels_attn.hip — fused attention core for the short sequences of this model (gfx950, wave64).
//
//   o = softmax(rope(q) rope(k)^T / sqrt(dh), keys j >= klen[b] excluded) v        per (batch b, head h)
//
// Sequences are tiny (text <= ~310 tokens, latent <= ~300 frames, 50 style tokens), so one workgroup owns a
// 32-query tile of one (b, h) and streams 64-key tiles of K and V through LDS with an online softmax.
// RoPE (plain or length-aware, arXiv:2509.11084) is applied while q / k are staged, so rotated copies
// never touch HBM.  fp32 math in both precisions; LDS rows are dh+1 words (odd stride: conflict-free).
// Thread (qi = tid / 8, g = tid % 8): scores for keys g + 8*jj, output columns g + 8*i of query row qi.
#include "kernels.hpp"
#include "kernels_dev.hpp"

#include <stdio.h>
#include <stdlib.h>

namespace stn {

static constexpr int AQ = 32, AK = 64, ADH_MAX = 96;

__device__ __forceinline__ float ld_act(const float* p) { return *p; }
__device__ __forceinline__ float ld_act(const uint16_t* p) { return __uint_as_float(((unsigned)*p) << 16); }
__device__ __forceinline__ float ld_act(const f16_t* p) { return (float)*p; }
__device__ __forceinline__ void st_act(f16_t* p, float v) { *p = (f16_t)v; }
__device__ __forceinline__ void st_act(float* p, float v) { *p = v; }
__device__ __forceinline__ void st_act(uint16_t* p, float v) {
    // round-to-nearest-even; inputs are finite softmax averages
    const unsigned u = __float_as_uint(v);
    *p = (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// stage `rows` rows (starting at sequence position pos0 of batch b) of one head into LDS, rotating pairs
// (i, i + dh/2) when rope_mode >= 0; rows at positions >= limit are zero-filled.
// fp32 rows: 16-byte loads, ALL of them issued before the first is consumed.  (Element by element this staging is a chain of
// rows*dh/512 dependent 4-byte loads per tensor and key tile — 12 for a 64 x 96 tile — and was 3/4 of the kernel's 63 us on a
// single utterance.)  Needs dh % 8 == 0, ld % 4 == 0 and 16-byte aligned rows; at most AK rows.
__device__ __forceinline__ void stage_rows_f32v(const float* __restrict__ src, int ld, int64_t seq_base, int pos0, int rows,
                                                int limit, int dh, int ds, float* __restrict__ dst, int rope_mode,
                                                float log_base, float gamma, int seq_len, float mul) {
    constexpr int MAXIT = (AK * (ADH_MAX / 8) + 255) / 256;  // 16-byte groups of half a row: rows * dh/8 of them, 256 threads
    const int hd2 = dh >> 1, q4 = hd2 >> 2, items = rows * q4;
    float4 a[MAXIT], b[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int idx = (int)threadIdx.x + it * 256;
        const int r = idx / q4, c = idx - r * q4;
        a[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        b[it] = a[it];
        if (idx < items && pos0 + r < limit) {
            const float* p = src + (seq_base + pos0 + r) * ld + 4 * c;
            a[it] = *reinterpret_cast<const float4*>(p);
            b[it] = *reinterpret_cast<const float4*>(p + hd2);
        }
    }
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int idx = (int)threadIdx.x + it * 256;
        if (idx >= items) continue;
        const int r = idx / q4, c = idx - r * q4;
        const int pos = pos0 + r;
        float x0[4] = {a[it].x, a[it].y, a[it].z, a[it].w}, x1[4] = {b[it].x, b[it].y, b[it].z, b[it].w};
        if (rope_mode >= 0 && pos < limit) {
            const float pp = rope_mode == 1 ? gamma * (float)pos / (float)(seq_len > 0 ? seq_len : 1) : (float)pos;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float inv = expf(-log_base * (float)(2 * (4 * c + e)) / (float)dh);
                float sn, cs;
                sincosf(pp * inv, &sn, &cs);
                const float a0 = x0[e], a1 = x1[e];
                x0[e] = a0 * cs - a1 * sn;
                x1[e] = a1 * cs + a0 * sn;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dst[r * ds + 4 * c + e] = x0[e] * mul;
            dst[r * ds + 4 * c + e + hd2] = x1[e] * mul;
        }
    }
}

template <typename T>
__device__ __forceinline__ void stage_rows(const T* __restrict__ src, int ld, int64_t seq_base, int pos0, int rows,
                                           int limit, int dh, int ds, float* __restrict__ dst, int rope_mode,
                                           float log_base, float gamma, int seq_len, float mul) {
    if constexpr (sizeof(T) == 4) {
        if (dh % 8 == 0 && ld % 4 == 0 && rows <= AK && blockDim.x == 256 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
            stage_rows_f32v(reinterpret_cast<const float*>(src), ld, seq_base, pos0, rows, limit, dh, ds, dst, rope_mode, log_base, gamma, seq_len, mul);
            return;
        }
    }
    const int hd2 = dh >> 1;
    for (int idx = threadIdx.x; idx < rows * hd2; idx += blockDim.x) {
        const int r = idx / hd2, i = idx - r * hd2;
        const int pos = pos0 + r;
        float x0 = 0.f, x1 = 0.f;
        if (pos < limit) {
            const T* p = src + (seq_base + pos) * ld;
            x0 = ld_act(p + i);
            x1 = ld_act(p + i + hd2);
            if (rope_mode >= 0) {
                const float pp = rope_mode == 1 ? gamma * (float)pos / (float)(seq_len > 0 ? seq_len : 1) : (float)pos;
                const float inv = expf(-log_base * (float)(2 * i) / (float)dh);
                float sn, cs;
                sincosf(pp * inv, &sn, &cs);
                const float a0 = x0, a1 = x1;
                x0 = a0 * cs - a1 * sn;
                x1 = a1 * cs + a0 * sn;
            }
        }
        dst[r * ds + i] = x0 * mul;
        dst[r * ds + i + hd2] = x1 * mul;
    }
}

// TPR = threads per query row: 8 (32 query rows per workgroup, 8 keys of a 64-key tile and dh/8 output columns per thread) or 32
// (8 query rows per workgroup, 2 keys and dh/32 columns per thread).  The work per thread — hence the latency of a launch that
// cannot fill the chip anyway (a single utterance: 8 workgroups at TPR = 8) — shrinks 4x with TPR = 32, at the price of four times
// as many workgroups staging the same keys; the launcher takes 32 when the grid would otherwise leave most CUs idle.
template <typename T, int TPR>
__global__ __launch_bounds__(256) void attn_kernel(const T* __restrict__ q, int ldq, const T* __restrict__ k,
                                                   const T* __restrict__ v, int ldk, T* __restrict__ o, int ldo, int Lq,
                                                   int Lk, int dh, const int* __restrict__ qlen,
                                                   const int* __restrict__ klen, int rope_mode, float log_base,
                                                   float gamma, int k_rot, const int* __restrict__ q_off,
                                                   const int* __restrict__ k_off) {
    constexpr int QR = 256 / TPR;        // query rows per workgroup
    constexpr int KPT = AK / TPR;        // keys per thread and tile
    constexpr int OPT = (ADH_MAX + TPR - 1) / TPR;  // output columns per thread
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ds = dh + 1;
    float* Qs = lds;               // [QR][ds]
    float* Ks = Qs + QR * ds;      // [AK][ds]
    float* Vs = Ks + AK * ds;      // [AK][ds]
    float* Ss = Vs + AK * ds;      // [QR][AK + 1]
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QR;
    const int tid = threadIdx.x, qi = tid / TPR, g = tid % TPR;
    const int nk = klen ? min(klen[b], Lk) : Lk;
    const int nq = qlen ? qlen[b] : Lq;  // used for length-aware positions only
    const float sc = rsqrtf((float)dh);

    const int64_t qrow0 = q_off ? (int64_t)q_off[b] : (int64_t)b * Lq;  // packed: the sequence owns nq rows from q_off[b]
    const int qrows = q_off ? nq : Lq;
    if (q0 >= qrows) return;  // uniform: nothing of this tile exists
    stage_rows<T>(q + h * dh, ldq, qrow0, q0, QR, qrows, dh, ds, Qs, rope_mode, log_base, gamma, nq, sc);

    float m_run = -1e30f, l_run = 0.f;
    float oacc[OPT];
#pragma unroll
    for (int i = 0; i < OPT; ++i) oacc[i] = 0.f;

    for (int k0 = 0; k0 < nk; k0 += AK) {
        __syncthreads();  // previous tile fully consumed (and Qs visible on the first pass)
        const int64_t krow0 = k_off ? (int64_t)k_off[b] : (int64_t)b * Lk;
        stage_rows<T>(k + h * dh, ldk, krow0, k0, AK, nk, dh, ds, Ks, k_rot ? -1 : rope_mode, log_base, gamma, nk, 1.f);
        stage_rows<T>(v + h * dh, ldk, krow0, k0, AK, nk, dh, ds, Vs, -1, 0.f, 0.f, 1, 1.f);
        __syncthreads();
        float s[KPT];
#pragma unroll
        for (int jj = 0; jj < KPT; ++jj) s[jj] = 0.f;
        const float* qrow = Qs + qi * ds;
        for (int d = 0; d < dh; ++d) {
            const float qv = qrow[d];
#pragma unroll
            for (int jj = 0; jj < KPT; ++jj) s[jj] = fmaf(qv, Ks[(g + TPR * jj) * ds + d], s[jj]);
        }
        float mx = -1e30f;
#pragma unroll
        for (int jj = 0; jj < KPT; ++jj) {
            if (k0 + g + TPR * jj >= nk) s[jj] = -1e30f;
            mx = fmaxf(mx, s[jj]);
        }
#pragma unroll
        for (int sh = 1; sh < TPR; sh <<= 1) mx = fmaxf(mx, __shfl_xor(mx, sh, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = expf(m_run - m_new);
        float ps = 0.f;
#pragma unroll
        for (int jj = 0; jj < KPT; ++jj) {
            const float p = s[jj] <= -1e29f ? 0.f : expf(s[jj] - m_new);
            ps += p;
            Ss[qi * (AK + 1) + g + TPR * jj] = p;
        }
#pragma unroll
        for (int sh = 1; sh < TPR; sh <<= 1) ps += __shfl_xor(ps, sh, 64);
        l_run = l_run * alpha + ps;
        m_run = m_new;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < OPT; ++i) oacc[i] *= alpha;
        const float* prow = Ss + qi * (AK + 1);
        for (int j = 0; j < AK; ++j) {
            const float p = prow[j];
            const float* vrow = Vs + j * ds + g;
#pragma unroll
            for (int i = 0; i < OPT; ++i)
                if (TPR * i + g < dh) oacc[i] = fmaf(p, vrow[TPR * i], oacc[i]);
        }
    }
    const int gq = q0 + qi;
    if (gq < qrows) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        T* orow = o + (qrow0 + gq) * ldo + h * dh + g;
#pragma unroll
        for (int i = 0; i < OPT; ++i)
            if (TPR * i + g < dh) st_act(orow + TPR * i, oacc[i] * inv);
    }
}


// ---------------------------------------------------------------------------------------------
// bf16 MFMA attention (v_mfma_f32_32x32x16_bf16), head dim DH in {32, 64, 96}.
//   workgroup = 4 waves = 128 queries of one (b, h); each wave owns 32 queries.
//   LDS: Q [128][DH] and a chunk of K [kc <= 128][DH] as bf16 rows of DH*2+16 bytes (the +16 makes every ds_read_b128
//   16-lane group hit 16 distinct slots), V TRANSPOSED [DH][lk_pad] with rows of lk_pad*2+8 bytes
//   (conflict-free ds_read_b64 per 32-lane half).  RoPE + 1/sqrt(dh)*log2(e) are applied while staging.
//   S^T = K Q^T is computed with the KEY on the accumulator rows and the QUERY on the lane, so the row max and
//   the row sum are 16 in-register ops + one cross-half shuffle.  The exponentiated tile is then the A operand
//   of O += P V with no lane movement (registers 8s..8s+7 = k-step s; element j <-> key 16s+8(j>>2)+4h+(j&3),
//   which is exactly the order the transposed V image is read in).
//   Softmax is two-pass (max, then exp/sum/PV with QK^T recomputed): sequences are <= ~320 keys, recomputing
//   the small QK^T is cheaper than rescaling O (whose rows live in registers, not on the lane).  Contexts longer than
//   one 128-key chunk stream K through the chunk buffer twice (once per pass) and V once.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    const unsigned ra = (ua + 0x7FFFu + ((ua >> 16) & 1u)) >> 16, rb = (ub + 0x7FFFu + ((ub >> 16) & 1u)) >> 16;
    return ra | (rb << 16);
}
// 16-bit storage format of the MFMA kernel: bf16 (F16 = false) or IEEE half (F16 = true); same layouts, same instruction timing
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
template <bool F16>
__device__ __forceinline__ unsigned pack_h2(float a, float b) {
    if constexpr (F16) { const f16x2_t h = {(_Float16)a, (_Float16)b}; return __builtin_bit_cast(unsigned, h); }
    else return pack_bf16x2(a, b);
}
template <bool F16>
__device__ __forceinline__ void unpack_h2(unsigned w, float& lo, float& hi) {
    if constexpr (F16) { const f16x2_t h = __builtin_bit_cast(f16x2_t, w); lo = (float)h[0]; hi = (float)h[1]; }
    else { lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xFFFF0000u); }
}
template <bool F16>
__device__ __forceinline__ f32x16_t mfma_h(bf16x8_t a, bf16x8_t b, f32x16_t c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// In-kernel phase stamps of one workgroup (measurement builds only: make EXTRA=-DSTN_ATTN_STAMPS; tools/attn_phases.py)
#ifdef STN_ATTN_STAMPS
__device__ unsigned long long g_attn_ts[8];
extern "C" void stn_dbg_attn_ts(unsigned long long* out) { (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_ts), sizeof(unsigned long long) * 8); }
#define ATTN_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 1 && blockIdx.z == 5 && threadIdx.x == 0) g_attn_ts[i] = __builtin_readcyclecounter(); } while (0)
#else
#define ATTN_STAMP(i) do { } while (0)
#endif
template <int DH, bool F16>
__global__ __launch_bounds__(256, 2) void attn_mfma_kernel(const uint16_t* __restrict__ q, int ldq,
                                                        const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
                                                        int ldk, uint16_t* __restrict__ o, int ldo, int Lq, int Lk,
                                                        int kc /* keys per LDS chunk: multiple of 32, <= 128 */,
                                                        const int* __restrict__ qlen,
                                                        const int* __restrict__ klen, int rope_mode, float log_base,
                                                        float gamma, int k_rot, const int* __restrict__ q_off,
                                                        const int* __restrict__ k_off) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ float inv_rev[DH / 2];  // rotation frequency of pair i in REVOLUTIONS per position unit (v_sin/v_cos input)
    constexpr int QS = DH * 2 + 16;  // bytes per Q / K row
    const int VS = kc * 2 + 8;       // bytes per V^T row
    unsigned char* Qs = lds_raw;
    unsigned char* Ks = Qs + 128 * QS;
    unsigned char* Vt = Ks + kc * QS;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nk = klen ? min(klen[b], Lk) : Lk;
    const int nq = qlen ? qlen[b] : Lq;
    const int64_t qrow0 = q_off ? (int64_t)q_off[b] : (int64_t)b * Lq;  // packed: the sequence owns nq rows from q_off[b]
    const int qrows = q_off ? nq : Lq;
    if (q0 >= qrows) return;  // uniform: nothing of this tile exists
    const int64_t krow0 = k_off ? (int64_t)k_off[b] : (int64_t)b * Lk;  // packed keys: the sequence owns nk rows from k_off[b]
    constexpr int HD2 = DH / 2;
    const float qmul = rsqrtf((float)DH) * 1.44269504088896340736f;

    ATTN_STAMP(0);
    if (rope_mode >= 0) {
        for (int i = tid; i < HD2; i += 256) inv_rev[i] = __expf(-log_base * (float)(2 * i) / (float)DH) * 0.15915494309189535f;
        __syncthreads();
    }
    constexpr int CH = HD2 / 8;  // 16-byte chunks per half row
    // ---- staging of Q (pass 0: 128 rows from q0) or of a chunk of K (pass 1: kc rows from key c0): RoPE in fp32, stored
    // 16-bit; 8 pairs per item, 16-byte loads and LDS stores.  A thread owns items tid, tid + 256, ...: ALL their loads are issued
    // first (unconditional buffer loads: an item outside the tile or the sequence reads zeros through the range check), the
    // arithmetic and the LDS stores follow — one memory round trip per staging step instead of one per item.  With one key chunk
    // (the estimator's cross-attention) the loads of Q, K and V are all in flight before the first store.
    constexpr int RN = (128 * CH + 255) / 256;        // items per thread of a 128-row tile
    constexpr int VN = (128 * (DH / 8) + 255) / 256;  // items per thread of a 128-key chunk of V
    const __amdgpu_buffer_rsrc_t rs_q = make_rsrc(q, 0x7FFFFFFFu), rs_k = make_rsrc(k, 0x7FFFFFFFu), rs_v = make_rsrc(v, 0x7FFFFFFFu);
    auto issue_rows = [&](int pass, int c0, u32x4_t (&w0)[RN], u32x4_t (&w1)[RN]) __attribute__((always_inline)) {
        const int ld = pass == 0 ? ldq : ldk;
        const int64_t seq_base = pass == 0 ? qrow0 : krow0;
        const int pos0 = pass == 0 ? q0 : c0, rows = pass == 0 ? 128 : kc, limit = pass == 0 ? qrows : nk;
#pragma unroll
        for (int i = 0; i < RN; ++i) {
            const int idx = tid + 256 * i, r = idx / CH, c = idx - r * CH, pos = pos0 + r;
            const unsigned off = (idx < rows * CH && pos < limit) ? (unsigned)(((seq_base + pos) * ld + h * DH + c * 8) * 2) : OOB;
            w0[i] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(pass == 0 ? rs_q : rs_k, off, 0, 0));
            w1[i] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(pass == 0 ? rs_q : rs_k, off, HD2 * 2, 0));
        }
    };
    auto commit_rows = [&](int pass, int c0, const u32x4_t (&w0)[RN], const u32x4_t (&w1)[RN]) __attribute__((always_inline)) {
        const int pos0 = pass == 0 ? q0 : c0, rows = pass == 0 ? 128 : kc;
        const int seq_len = pass == 0 ? nq : nk;
        const float mul = pass == 0 ? qmul : 1.f;
        const bool rot = rope_mode >= 0 && !(pass == 1 && k_rot);
        const float pscale = rope_mode == 1 ? gamma / (float)(seq_len > 0 ? seq_len : 1) : 1.f;
        unsigned char* dst = pass == 0 ? Qs : Ks;
#pragma unroll
        for (int i = 0; i < RN; ++i) {
            const int idx = tid + 256 * i, r = idx / CH, c = idx - r * CH, pos = pos0 + r;
            if (idx >= rows * CH) continue;
            u32x4_t o0, o1;
            if (rot || mul != 1.f) {
                const float pp = (float)pos * pscale;
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    float a0[2], a1[2];
                    unpack_h2<F16>(w0[i][e2], a0[0], a0[1]);
                    unpack_h2<F16>(w1[i][e2], a1[0], a1[1]);
                    float y0[2], y1[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        float cs = 1.f, sn = 0.f;
                        if (rot) {
                            const float rev = __builtin_amdgcn_fractf(pp * inv_rev[c * 8 + 2 * e2 + u]);
                            sn = __builtin_amdgcn_sinf(rev);
                            cs = __builtin_amdgcn_cosf(rev);
                        }
                        y0[u] = (a0[u] * cs - a1[u] * sn) * mul;
                        y1[u] = (a1[u] * cs + a0[u] * sn) * mul;
                    }
                    o0[e2] = pack_h2<F16>(y0[0], y0[1]);
                    o1[e2] = pack_h2<F16>(y1[0], y1[1]);
                }
            } else {
                o0 = w0[i]; o1 = w1[i];
            }
            *reinterpret_cast<u32x4_t*>(dst + r * QS + c * 16) = o0;
            *reinterpret_cast<u32x4_t*>(dst + r * QS + HD2 * 2 + c * 16) = o1;
        }
    };
    // ---- a chunk of V, transposed ---------------------------------------------------------------------------------------
    auto issue_v = [&](int c0, u32x4_t (&w)[VN]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < VN; ++i) {
            const int idx = tid + 256 * i, kl = idx / (DH / 8), c = idx - kl * (DH / 8), key = c0 + kl;
            const unsigned off = (idx < kc * (DH / 8) && key < nk) ? (unsigned)(((krow0 + key) * ldk + h * DH + c * 8) * 2) : OOB;
            w[i] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_v, off, 0, 0));
        }
    };
    auto commit_v = [&](const u32x4_t (&w)[VN]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < VN; ++i) {
            const int idx = tid + 256 * i, kl = idx / (DH / 8), c = idx - kl * (DH / 8);
            if (idx >= kc * (DH / 8)) continue;
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                const unsigned word = w[i][e2];
                *reinterpret_cast<uint16_t*>(Vt + (c * 8 + 2 * e2) * VS + kl * 2) = (uint16_t)(word & 0xFFFFu);
                *reinterpret_cast<uint16_t*>(Vt + (c * 8 + 2 * e2 + 1) * VS + kl * 2) = (uint16_t)(word >> 16);
            }
        }
    };
    auto stage_rows = [&](int pass, int c0) __attribute__((always_inline)) {
        u32x4_t w0[RN], w1[RN];
        issue_rows(pass, c0, w0, w1);
        __builtin_amdgcn_sched_barrier(0);
        commit_rows(pass, c0, w0, w1);
    };
    auto stage_v = [&](int c0) __attribute__((always_inline)) {
        u32x4_t w[VN];
        issue_v(c0, w);
        __builtin_amdgcn_sched_barrier(0);
        commit_v(w);
    };

    const int qbase = wave * 32;
    const bool active = q0 + qbase < qrows;  // waves whose 32 queries are all padding still stage and keep the barriers
    const int lr = lane & 31, lh = lane >> 5;
    const int nkt = kc >> 5;
    const int nch = nk > 0 ? (nk + kc - 1) / kc : 1;  // key chunks (uniform per workgroup)
    bf16x8_t bq[DH / 16];
    float m = -1e30f, lsum = 0.f;
    f32x16_t oacc[DH / 32];
#pragma unroll
    for (int nd = 0; nd < DH / 32; ++nd)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[nd][i] = 0.f;

    auto load_q = [&]() {
#pragma unroll
        for (int ks = 0; ks < DH / 16; ++ks)
            bq[ks] = *reinterpret_cast<const bf16x8_t*>(Qs + (qbase + lr) * QS + (ks * 2 + lh) * 16);
    };
    auto pass_max = [&](int c0) {  // running row maximum over the staged chunk
        for (int kt = 0; kt < nkt; ++kt) {
            if (c0 + kt * 32 >= nk) break;
            f32x16_t acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < DH / 16; ++ks) {
                const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(Ks + (kt * 32 + lr) * QS + (ks * 2 + lh) * 16);
                acc = mfma_h<F16>(a, bq[ks], acc);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = c0 + kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
                m = fmaxf(m, key < nk ? acc[i] : -1e30f);
            }
        }
    };
    auto pass_pv = [&](int c0) {  // exp, row sums and P V over the staged chunk (QK^T recomputed)
        for (int kt = 0; kt < nkt; ++kt) {
            if (c0 + kt * 32 >= nk) break;
            f32x16_t acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < DH / 16; ++ks) {
                const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(Ks + (kt * 32 + lr) * QS + (ks * 2 + lh) * 16);
                acc = mfma_h<F16>(a, bq[ks], acc);
            }
            float p[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = c0 + kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
                p[i] = key < nk ? exp2f(acc[i] - m) : 0.f;
                lsum += p[i];
            }
#pragma unroll
            for (int sidx = 0; sidx < 2; ++sidx) {
                u32x4_t pw;
                pw[0] = pack_h2<F16>(p[8 * sidx + 0], p[8 * sidx + 1]);
                pw[1] = pack_h2<F16>(p[8 * sidx + 2], p[8 * sidx + 3]);
                pw[2] = pack_h2<F16>(p[8 * sidx + 4], p[8 * sidx + 5]);
                pw[3] = pack_h2<F16>(p[8 * sidx + 6], p[8 * sidx + 7]);
                const bf16x8_t ap = __builtin_bit_cast(bf16x8_t, pw);
#pragma unroll
                for (int nd = 0; nd < DH / 32; ++nd) {
                    const unsigned char* base = Vt + (nd * 32 + lr) * VS + (kt * 32 + 16 * sidx + 4 * lh) * 2;
                    const uint2 lo = *reinterpret_cast<const uint2*>(base);
                    const uint2 hi = *reinterpret_cast<const uint2*>(base + 16);
                    u32x4_t vw;
                    vw[0] = lo.x; vw[1] = lo.y; vw[2] = hi.x; vw[3] = hi.y;
                    oacc[nd] = mfma_h<F16>(ap, __builtin_bit_cast(bf16x8_t, vw), oacc[nd]);
                }
            }
        }
    };

    if (nch == 1) {  // the whole context fits one chunk: K is staged once and serves both passes
        u32x4_t qa[RN], qb[RN], ka[RN], kb[RN], va[VN];
        issue_rows(0, 0, qa, qb);
        issue_rows(1, 0, ka, kb);
        issue_v(0, va);
        __builtin_amdgcn_sched_barrier(0);
        commit_rows(0, 0, qa, qb);
        ATTN_STAMP(1);
        commit_rows(1, 0, ka, kb);
        ATTN_STAMP(2);
        commit_v(va);
        __syncthreads();
        ATTN_STAMP(3);
        if (active) {
            load_q();
            pass_max(0);
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            ATTN_STAMP(4);
            pass_pv(0);
        }
    } else {  // long contexts (up to ~320 text tokens): K streams through the chunk buffer twice, V once
        stage_rows(0, 0);
        for (int c = 0; c < nch; ++c) {
            if (c) __syncthreads();  // every wave is done with the previous chunk
            stage_rows(1, c * kc);
            __syncthreads();
            if (active) {
                if (c == 0) load_q();
                pass_max(c * kc);
            }
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        for (int c = 0; c < nch; ++c) {
            __syncthreads();
            stage_rows(1, c * kc);
            stage_v(c * kc);
            __syncthreads();
            if (active) pass_pv(c * kc);
        }
    }
    // Wait states between the last P V MFMA and the first read of its accumulators.  The compiler pads such reads with s_nop
    // inside a block but not on this loop-exit edge (ROCm 7.2: v_mfma a[0:15] / s_cbranch / 4 scalar ops / v_accvgpr_read a15,
    // 6 wait states where 11 are required): query rows 27 and 31 of a wave came back without the last key tile's contribution
    // (fp16 build, contexts of 49-64 keys).  The accumulators are operands of the asm, so every read is ordered behind it;
    // tools/check_hazards.py (tests/test_isa_cpu.py) walks the shipped code objects for this pattern.
    if constexpr (DH == 32) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(oacc[0]));
    else if constexpr (DH == 64) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(oacc[0]), "+a"(oacc[1]));
    else asm volatile("s_nop 15\n\ts_nop 7" : "+a"(oacc[0]), "+a"(oacc[1]), "+a"(oacc[2]));
    ATTN_STAMP(5);
    if (!active) return;
    lsum += __shfl_xor(lsum, 32, 64);
    const float inv = (nk > 0 && lsum > 0.f) ? 1.0f / lsum : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int qrow = (i & 3) + 8 * (i >> 2) + 4 * lh;
        const float invq = __shfl(inv, qrow, 64);  // lane qrow (< 32) holds the sum of query qrow
        const int gq = q0 + qbase + qrow;
        if (gq < qrows) {
            uint16_t* orow = o + (qrow0 + gq) * ldo + h * DH + lr;
#pragma unroll
            for (int nd = 0; nd < DH / 32; ++nd) orow[nd * 32] = (uint16_t)pack_h2<F16>(oacc[nd][i] * invq, 0.f);
        }
    }
#ifdef STN_ATTN_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    ATTN_STAMP(6);
#endif
}

template <int DH, bool F16>
static void launch_attn_mfma(hipStream_t s, const uint16_t* q, int ldq, const uint16_t* k, const uint16_t* v, int ldk,
                             uint16_t* o, int ldo, int B, int Lq, int Lk, int H, int kc, size_t lds, const int* qlen,
                             const int* klen, int rope_mode, float log_base, float gamma, int k_rot, const int* q_off, const int* k_off) {
    static PerDeviceOnce attr_once;
    if (attr_once.need()) {
        // 150 KiB dynamic (the launcher's own bound) + the kernel's static table stay inside the CU's 160 KiB
        stn_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma_kernel<DH, F16>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024), "hipFuncSetAttribute(attn_mfma)");
    }
    const dim3 grid((Lq + 127) / 128, H, B);
    STN_KLAUNCH((attn_mfma_kernel<DH, F16>), grid, dim3(256), lds, s, q, ldq, k, v, ldk, o, ldo, Lq, Lk, kc, qlen, klen,
                       rope_mode, log_base, gamma, k_rot, q_off, k_off);
}

void launch_attention(hipStream_t s, int dtype, const void* q, int ldq, const void* k, const void* v, int ldk, void* o,
                      int ldo, int B, int Lq, int Lk, int H, int dh, const int* qlen, const int* klen, int rope_mode,
                      float rope_base, float rope_gamma, bool k_rotated, const int* q_off, const int* k_off) {
    if (B == 0 || Lq == 0) return;
    if ((q_off && !qlen) || (k_off && !klen)) { throw std::invalid_argument("packed attention needs the lengths of the packed side"); }
    if (dh > ADH_MAX || dh % 8 || dh < 8) { char m_[256]; snprintf(m_, sizeof m_, "attention head dim %d unsupported (multiple of 8, <= %d)", dh, ADH_MAX); throw std::invalid_argument(m_); }
    // (the MFMA kernel addresses q, k and v through 32-bit buffer offsets: operands of 2 GiB and more take the scalar kernel)
    if (is_half(dtype) && (dh == 32 || dh == 64 || dh == 96) && ldk % 8 == 0 && ldq % 8 == 0 && !(reinterpret_cast<uintptr_t>(v) & 15) &&
        !(reinterpret_cast<uintptr_t>(q) & 15) && !(reinterpret_cast<uintptr_t>(k) & 15) && (int64_t)B * Lq * ldq * 2 < 0x7FFFFFFFll &&
        (int64_t)B * Lk * ldk * 2 < 0x7FFFFFFFll) {
        // keys go through LDS in chunks of at most 128 (one chunk covers the 50 style tokens and ~100-token texts; longer texts
        // take several), so the MFMA kernel serves every context length at 2 workgroups per CU
        const int lk_pad = (Lk + 31) & ~31;
        const int kc = lk_pad < 128 ? lk_pad : 128;
        const size_t need = (size_t)128 * (dh * 2 + 16) + (size_t)kc * (dh * 2 + 16) + (size_t)dh * (kc * 2 + 8);
        const uint16_t *q16 = static_cast<const uint16_t*>(q), *k16 = static_cast<const uint16_t*>(k), *v16 = static_cast<const uint16_t*>(v);
        uint16_t* o16 = static_cast<uint16_t*>(o);
        const float lb = logf(rope_base);
        if (dh == 32) { if (dtype == F16) launch_attn_mfma<32, true>(s, q16, ldq, k16, v16, ldk, o16, ldo, B, Lq, Lk, H, kc, need, qlen, klen, rope_mode, lb, rope_gamma, (int)k_rotated, q_off, k_off); else launch_attn_mfma<32, false>(s, q16, ldq, k16, v16, ldk, o16, ldo, B, Lq, Lk, H, kc, need, qlen, klen, rope_mode, lb, rope_gamma, (int)k_rotated, q_off, k_off); }
        else if (dh == 64) { if (dtype == F16) launch_attn_mfma<64, true>(s, q16, ldq, k16, v16, ldk, o16, ldo, B, Lq, Lk, H, kc, need, qlen, klen, rope_mode, lb, rope_gamma, (int)k_rotated, q_off, k_off); else launch_attn_mfma<64, false>(s, q16, ldq, k16, v16, ldk, o16, ldo, B, Lq, Lk, H, kc, need, qlen, klen, rope_mode, lb, rope_gamma, (int)k_rotated, q_off, k_off); }
        else { if (dtype == F16) launch_attn_mfma<96, true>(s, q16, ldq, k16, v16, ldk, o16, ldo, B, Lq, Lk, H, kc, need, qlen, klen, rope_mode, lb, rope_gamma, (int)k_rotated, q_off, k_off); else launch_attn_mfma<96, false>(s, q16, ldq, k16, v16, ldk, o16, ldo, B, Lq, Lk, H, kc, need, qlen, klen, rope_mode, lb, rope_gamma, (int)k_rotated, q_off, k_off); }
        return;
    }
    const int ds = dh + 1;
    const float log_base = logf(rope_base);
    // few workgroups (single utterances): 32 threads per query row cut the work per thread, i.e. the launch's latency, by four
    const bool wide = (int64_t)((Lq + AQ - 1) / AQ) * H * B < 96;
    const int qr = wide ? 8 : AQ;
    const size_t lds = sizeof(float) * ((size_t)(qr + 2 * AK) * ds + (size_t)qr * (AK + 1));
    const dim3 grid((Lq + qr - 1) / qr, H, B);
    static PerDeviceOnce attr_once;
    if (attr_once.need()) {
        stn_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<float, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024), "hipFuncSetAttribute(attn f32)");
        stn_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<uint16_t, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024), "hipFuncSetAttribute(attn bf16)");
        stn_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<f16_t, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024), "hipFuncSetAttribute(attn f16)");
        stn_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<float, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024), "hipFuncSetAttribute(attn f32 wide)");
        stn_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<uint16_t, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024), "hipFuncSetAttribute(attn bf16 wide)");
        stn_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<f16_t, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024), "hipFuncSetAttribute(attn f16 wide)");
    }
#define STN_ATTN_GO(T_, TPR_)                                                                                                  \
    STN_KLAUNCH((attn_kernel<T_, TPR_>), grid, dim3(256), lds, s, static_cast<const T_*>(q), ldq, static_cast<const T_*>(k),      \
                static_cast<const T_*>(v), ldk, static_cast<T_*>(o), ldo, Lq, Lk, dh, qlen, klen, rope_mode, log_base, rope_gamma, \
                (int)k_rotated, q_off, k_off)
    if (dtype == F16) { if (wide) STN_ATTN_GO(f16_t, 32); else STN_ATTN_GO(f16_t, 8); }
    else if (dtype == BF16) { if (wide) STN_ATTN_GO(uint16_t, 32); else STN_ATTN_GO(uint16_t, 8); }
    else { if (wide) STN_ATTN_GO(float, 32); else STN_ATTN_GO(float, 8); }
#undef STN_ATTN_GO
}

// ---------------------------------------------------------------------------------------------
// one-time key rotation (see kernels.hpp)
// ---------------------------------------------------------------------------------------------
// A thread owns one (row, rotation pair i): the angle depends on nothing else, so exp / sincos run once and serve every group (block) and head of the
// row (the former one-thread-per-element form computed them groups x H = 16 times over: 43.6 us at the bench's shape, this one 12)
template <typename T>
__global__ void rope_rows_kernel(T* __restrict__ x, int ld, int L, const int* __restrict__ len, int groups, int group_stride,
                                 int H, int dh, int rope_mode, float log_base, float gamma, int64_t n,
                                 const int* __restrict__ row_off) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over [B*L][dh/2]
    if (idx >= n) return;
    const int hd2 = dh >> 1;
    const int i = (int)(idx % hd2);
    const int64_t r = idx / hd2;
    const int pos = (int)(r % L), b = (int)(r / L);
    const int nb = len ? min(len[b], L) : L;
    if (pos >= nb) return;
    const int64_t row = row_off ? (int64_t)row_off[b] + pos : r;  // packed rows: position pos of sequence b
    const float pp = rope_mode == 1 ? gamma * (float)pos / (float)(nb > 0 ? nb : 1) : (float)pos;
    const float inv = expf(-log_base * (float)(2 * i) / (float)dh);
    float sn, cs;
    sincosf(pp * inv, &sn, &cs);
    for (int g = 0; g < groups; ++g)
        for (int h = 0; h < H; ++h) {
            T* p = x + row * ld + (int64_t)g * group_stride + h * dh;
            const float a0 = ld_act(p + i), a1 = ld_act(p + i + hd2);
            st_act(p + i, a0 * cs - a1 * sn);
            st_act(p + i + hd2, a1 * cs + a0 * sn);
        }
}

void launch_rope_rows(hipStream_t s, int dtype, void* x, int ld, int B, int L, const int* len, int groups, int group_stride,
                      int H, int dh, int rope_mode, float rope_base, float rope_gamma, const int* row_off) {
    const int64_t n = (int64_t)B * L * (dh / 2);
    if (n == 0 || rope_mode < 0 || groups * H == 0) return;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (dtype == F16)
        STN_KLAUNCH(rope_rows_kernel<f16_t>, grid, dim3(256), 0, s, static_cast<f16_t*>(x), ld, L, len, groups,
                           group_stride, H, dh, rope_mode, logf(rope_base), rope_gamma, n, row_off);
    else if (dtype == BF16)
        STN_KLAUNCH(rope_rows_kernel<uint16_t>, grid, dim3(256), 0, s, static_cast<uint16_t*>(x), ld, L, len, groups,
                           group_stride, H, dh, rope_mode, logf(rope_base), rope_gamma, n, row_off);
    else
        STN_KLAUNCH(rope_rows_kernel<float>, grid, dim3(256), 0, s, static_cast<float*>(x), ld, L, len, groups, group_stride,
                           H, dh, rope_mode, logf(rope_base), rope_gamma, n, row_off);
}

}  // namespace stn
