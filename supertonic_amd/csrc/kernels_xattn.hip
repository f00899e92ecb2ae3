// kernels_xattn.hip — one launch per cross-attention block of the vector estimator (gfx950, wave64, bf16 / f16 modes):
//
//     x <- x + Wo . attention(q = Wq . LN(x) + bq,  K, V) + bo          for the rows one utterance owns
//
// The unfused block is four launches (LayerNorm — or fold_ln when the previous ConvNeXt block ran as K4-split —, q GEMM, attention,
// output GEMM + residual); 40 of them run per 128-utterance batch (4 blocks x text/style x 5 Euler steps), 14 % of the batch.  Here
// one workgroup (4 waves) owns a tile of QT = 32*MT query rows of one utterance and keeps everything between the two global accesses
// of x in LDS / registers:
//   0. (fold of a pending K4-split update +) LayerNorm of the tile's rows, one row per half wavefront, DPP reductions -> XS [QT][C]
//      (16-bit); the fp32 rows stay in registers (`keep`) until the final store
//   1. Q = XS . Wq^T: wave w owns output columns [w*C/4, (w+1)*C/4); A fragments from XS (ds_read_b128), B fragments straight
//      from global memory in the MFMA operand layout (lane = output column, 8 consecutive k = 16 B), XA_PD k-steps ahead in
//      registers; Q + bias is written back over XS as 16-bit (style blocks: scaled here; text blocks: rotated and scaled in place)
//   2. per head pair: K / V^T of the context staged in LDS (keys arrive already rotated), QK^T / softmax / PV as attn_mfma_kernel's
//      single-chunk path (keys on the accumulator rows); O_h overwrites Q_h in XS
//   3. Y = XS . Wo^T like step 1, Y + bo through an fp32 LDS image, x = keep + Y stored once
// Contexts longer than one 128-key chunk, other widths and fp32 engines take the unfused path (Engine::ve_step_dev).
//
// STATUS (round 3): correct (tests/test_gpu_xattn.py: equal to the four-launch form far inside one 16-bit rounding, both layouts, bf16
// and f16, with and without a pending fold) and, after a pass over every phase, AT PARITY with the four launches, not ahead: 49 us
// per block against 44 us for the four (which also cost three launch boundaries), 12.03 against 11.95 ms per batch on one box —
// so it stays opt-in (stn_set_fused_xattn / STN_XATTN=1).  In-kernel stamps of a 64-row workgroup (-DXA_STAMPS, tools/xattn_phases.py,
// profiles/r03_xattn_phases.txt), round 1 -> now, k cycles: fold + LayerNorm 62 -> 31 · q projection 13.6 -> 12.2 · q write-back +
// rotation 12.8 -> 5.1 · K/V staging 2 x 8 -> 2 x 5 · attention 2 x 8.7 -> 2 x 7.4 · output projection 15.9 -> 14.7 · residual 21 -> 8
// = 161 -> 99.  What bounds it is INGEST PER CU, not arithmetic and not scheduling: one workgroup per utterance means 128 workgroups
// that each pull their tile's x and four partial sums (295 KB) from beyond L2 at the ~13 B/clk a single CU gets from the Infinity
// Cache (31 k cycles whether the loads are issued 2, 4 or 8 rows deep, with or without the store of the folded rows), and each
// stream both weight matrices (2 x 295 KB) from L2 at ~25 B/clk (12 k cycles per projection for 4.6 k cycles of MFMA, whether 4 or 8
// k-steps are in flight, A fragments read ahead or not).  The four launches spread the same bytes over 256 CUs with 29 waves each.
// 32-row tiles (256 workgroups) halve phase 0 but double the weight traffic and leave half the waves idle in phase 2: 74 us.
#include "kernels.hpp"
#include "kernels_fold.hpp"

#include <algorithm>
#include <stdio.h>
#include <stdlib.h>

namespace stn {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
template <bool F16>
__device__ __forceinline__ unsigned pack_h2(float a, float b) {
    if constexpr (F16) { const f16x2_t h = {(_Float16)a, (_Float16)b}; return __builtin_bit_cast(unsigned, h); }
    else { const bf16x2_t h = {(__bf16)a, (__bf16)b}; return __builtin_bit_cast(unsigned, h); }  // v_cvt_pk_bf16_f32: one instruction (RNE), the
                                                                                                // integer form is ~6 per value at one wave per SIMD
}
template <bool F16>
__device__ __forceinline__ void unpack_h2(unsigned w, float& lo, float& hi) {
    if constexpr (F16) { const f16x2_t h = __builtin_bit_cast(f16x2_t, w); lo = (float)h[0]; hi = (float)h[1]; }
    else { lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xFFFF0000u); }
}
template <bool F16>
__device__ __forceinline__ float unround16(float v) {  // v rounded to the 16-bit storage format and back
    float lo, hi;
    unpack_h2<F16>(pack_h2<F16>(v, 0.f), lo, hi);
    return lo;
}
template <bool F16>
__device__ __forceinline__ f32x16_t mfma_h(bf16x8_t a, bf16x8_t b, f32x16_t c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// acc[mt][nt] += XS[32*mt .. +32][:] . W[n0 + 32*nt .. +32][:]^T over K = C.  The weight arrives in FRAGMENT ORDER
// (launch_repack_frag): the 64 x 16 bytes one MFMA B operand needs — lane (lr, lh) of n-tile T, k-step ks holds
// W[32*T + lr][16*ks + 8*lh .. +8] — are one contiguous KiB at ((T * C/16 + ks) * 64 + lane) * 16 bytes, so every load is a
// fully coalesced wave-wide KiB and consecutive k-steps are consecutive KiBs.  (Read straight from the row-major matrix the
// same loads touch 32 lines for 32 bytes each: measured 85 us per block instead of the four-launch form's 43.)
// XA_PD k-steps stay in flight in registers; `bfr` arrives pre-loaded with k-steps 0..XA_PD-1 (frag_prefetch), so that a caller can
// issue them ahead of unrelated work.
static constexpr int XA_SU = 5;  // K / V staging items whose loads are in flight together per thread
static constexpr int XA_PD = 8;  // k-steps of a weight matrix in flight per wavefront (registers): L2 latency under 128+ workgroups
                                 // reading the same matrix is ~2 k cycles, a k-step of six MFMAs 200
template <int C, int NTW>
__device__ __forceinline__ void frag_prefetch(const uint16_t* __restrict__ Wf, int n0, int lane, bf16x8_t (&bfr)[XA_PD][NTW]) {
    constexpr int NKS = C / 16;
#pragma unroll
    for (int p = 0; p < XA_PD; ++p)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
            bfr[p][nt] = *reinterpret_cast<const bf16x8_t*>(Wf + ((size_t)((n0 / 32 + nt) * NKS + p) * 64 + lane) * 8);
}
template <int C, int MT, int NTW, bool F16>
__device__ __forceinline__ void rows_times_wt(const unsigned char* __restrict__ XS, int XSTR, const uint16_t* __restrict__ Wf, int n0,
                                              int lane, bf16x8_t (&bfr)[XA_PD][NTW], f32x16_t (&acc)[MT][NTW]) {
    constexpr int NKS = C / 16, PD = XA_PD;
    static_assert(NKS % PD == 0, "K must be a multiple of 16 * XA_PD");
    const int lr = lane & 31, lh = lane >> 5;
    // the A fragments of a whole block of PD k-steps are read from LDS before its first MFMA (read one k-step at a time, each k-step
    // waits out the ~120 cycles of LDS latency: 24 x 120 cycles beside 24 x 192 of MFMA)
#pragma unroll
    for (int ks0 = 0; ks0 < NKS; ks0 += PD) {
        bf16x8_t a[PD][MT];
#pragma unroll
        for (int p = 0; p < PD; ++p)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[p][mt] = *reinterpret_cast<const bf16x8_t*>(XS + (mt * 32 + lr) * XSTR + ((ks0 + p) * 2 + lh) * 16);
#pragma unroll
        for (int p = 0; p < PD; ++p) {
            const int ks = ks0 + p;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = mfma_h<F16>(a[p][mt], bfr[p][nt], acc[mt][nt]);
            if (ks + PD < NKS) {
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    bfr[p][nt] = *reinterpret_cast<const bf16x8_t*>(Wf + ((size_t)((n0 / 32 + nt) * NKS + ks + PD) * 64 + lane) * 8);
            }
        }
    }
}

// Phase 0 of the fused block: rows [0, nrows) of the tile (global rows xrow0 ..) -> XS as 16-bit LayerNorm output.
// S > 0: a pending K4-split update is folded first (kernels_fold.hpp) and the folded rows are stored back to x.
template <int C, int MT, bool F16, int S, bool RV>
__device__ __forceinline__ void xa_ln_phase(float* __restrict__ x, int64_t xrow0, int nrows, unsigned char* __restrict__ XS, int XSTR,
                                            const float* __restrict__ ln_g, const float* __restrict__ ln_b, float eps,
                                            const uint16_t* __restrict__ fpart, int64_t fstride, const float* __restrict__ fb2,
                                            const float* __restrict__ fgamma, const float* __restrict__ frv, int tid,
                                            float4 (&keep)[MT * 4][C / 128]) {
    constexpr int C4 = C / 4, NS = C4 / 32, QT = MT * 32;
    static_assert(C4 % 32 == 0, "a half wavefront covers a row in whole float4 slots");
    constexpr int U = S > 4 ? 1 : (MT >= 2 ? 4 : 2);  // rows of a half wavefront in flight (one workgroup per CU: ~150 KB in flight saturate its share)
    const int lane = tid & 63, l32 = lane & 31, hw = lane >> 5, wave = tid >> 6;
    float4 fb[NS], fg[NS], ft[NS];
    if constexpr (S > 0) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            fb[i] = reinterpret_cast<const float4*>(fb2)[l32 + 32 * i];
            fg[i] = reinterpret_cast<const float4*>(fgamma)[l32 + 32 * i];
            ft[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (RV) ft[i] = reinterpret_cast<const float4*>(frv)[l32 + 32 * i];
        }
    }
    float4 gg[NS], bb[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) { gg[i] = reinterpret_cast<const float4*>(ln_g)[l32 + 32 * i]; bb[i] = reinterpret_cast<const float4*>(ln_b)[l32 + 32 * i]; }
    const float inv_c = 1.0f / (float)C;
    // half wavefront (wave, hw) owns rows 2 * wave + hw + 8 * k.  The (folded) fp32 rows stay in `keep` for the whole kernel: the
    // residual update at the end adds the block's output to them and stores x once — no store of the folded rows here, no second
    // read of x there (a third of the workgroup's memory traffic, which is what bounds this phase: ~13 B/clk per CU from the MALL).
#pragma unroll
    for (int k0 = 0; k0 < QT / 8; k0 += U) {
        float4 h[U][NS];
        // every load of the pass, then a scheduling barrier: left to itself hipcc sinks most of these loads behind the arithmetic of
        // the previous ones ("load, s_waitcnt vmcnt(0), use" one at a time: a global round trip per load instead of one per pass)
        constexpr int CH = S == 0 ? 1 : (S <= 12 ? S : 12);
        unsigned pw[U][NS][CH][2];
        int64_t mrow[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = 2 * wave + hw + 8 * (k0 + u);
            mrow[u] = xrow0 + (r < nrows ? r : 0);  // (rows past the end: row 0 again, nothing stored)
            const float4* x4 = reinterpret_cast<const float4*>(x + mrow[u] * C);
#pragma unroll
            for (int i = 0; i < NS; ++i) h[u][i] = x4[l32 + 32 * i];
            if constexpr (S > 0) {
#pragma unroll
                for (int i = 0; i < NS; ++i)
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const uint2 q = *reinterpret_cast<const uint2*>(fpart + (size_t)c * fstride + (size_t)mrow[u] * C + (l32 + 32 * i) * 4);
                        pw[u][i][c][0] = q.x; pw[u][i][c][1] = q.y;
                    }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (S > 0) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    float acc[4];
#pragma unroll
                    for (int c = 0; c < CH; ++c)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const float lo = p16_to_f<F16>(pw[u][i][c][j] & 0xFFFFu), hi = p16_to_f<F16>(pw[u][i][c][j] >> 16);
                            if (c == 0) { acc[2 * j] = lo; acc[2 * j + 1] = hi; }
                            else { acc[2 * j] += lo; acc[2 * j + 1] += hi; }
                        }
                    if constexpr (S > CH) {  // the remaining splits, chunk by chunk, in split order
#pragma unroll
                        for (int s0 = CH; s0 < S; s0 += CH) {
                            fold_chunk<F16, 4, CH, false>(fpart + (size_t)mrow[u] * C + (l32 + 32 * i) * 4, fstride, s0, acc);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    h[u][i] = fold_four(h[u][i], acc, fb[i], fg[i], ft[i]);
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = 2 * wave + hw + 8 * (k0 + u);
            const bool live = r < nrows;
#pragma unroll
            for (int i = 0; i < NS; ++i) keep[k0 + u][i] = h[u][i];
            float sm = 0.f;
#pragma unroll
            for (int i = 0; i < NS; ++i) sm += (h[u][i].x + h[u][i].y) + (h[u][i].z + h[u][i].w);
            const float mean = half_wave_sum(sm) * inv_c;
            float vr = 0.f;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const float dx = h[u][i].x - mean, dy = h[u][i].y - mean, dz = h[u][i].z - mean, dw = h[u][i].w - mean;
                vr += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
            const float rstd = rsqrtf(half_wave_sum(vr) * inv_c + eps);
            if (r < QT) {
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    uint2 o = make_uint2(0u, 0u);  // rows of the tile beyond the utterance: zeros (their MFMA rows are never stored)
                    if (live) {
                        o.x = pack_h2<F16>((h[u][i].x - mean) * rstd * gg[i].x + bb[i].x, (h[u][i].y - mean) * rstd * gg[i].y + bb[i].y);
                        o.y = pack_h2<F16>((h[u][i].z - mean) * rstd * gg[i].z + bb[i].z, (h[u][i].w - mean) * rstd * gg[i].w + bb[i].w);
                    }
                    *reinterpret_cast<uint2*>(XS + r * XSTR + (l32 + 32 * i) * 8) = o;
                }
            }
        }
    }
}

#ifdef XA_STAMPS
__device__ unsigned long long g_xa_ts[16];
extern "C" void stn_dbg_xa(unsigned long long* out) { (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xa_ts), sizeof(unsigned long long) * 16); }
#define XA_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 5 && threadIdx.x == 0) g_xa_ts[i] = __builtin_readcyclecounter(); } while (0)
#else
#define XA_STAMP(i) do { } while (0)
#endif
// PART 0: the whole block in one launch.  PART 1 / PART 2: the same phases as two launches cut behind the q projection — part 1 (fold +
// LayerNorm + q projection + rotation, 32-row tiles: twice the workgroups for the half of the block whose cost is rows) leaves the folded
// rows in x and the rotated, scaled q rows in `qbuf`; part 2 (attention + output projection + residual) reads them back.
template <int C, int DH, int MT, bool F16, int PART>
__global__ __launch_bounds__(256) void xattn_fused_kernel(float* __restrict__ x, const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                          float eps, const uint16_t* __restrict__ Wq, const float* __restrict__ bq,
                                                          const uint16_t* __restrict__ kp, const uint16_t* __restrict__ vp, int ldk,
                                                          const uint16_t* __restrict__ Wo, const float* __restrict__ bo, int L, int Lk,
                                                          int kc /* keys in the LDS chunk: multiple of 32, <= 128, >= Lk */,
                                                          const int* __restrict__ qlen, const int* __restrict__ klen,
                                                          const int* __restrict__ q_off, const int* __restrict__ k_off, int rope_mode,
                                                          float log_base, float gamma,
                                                          const uint16_t* __restrict__ fpart, int fS, int64_t fstride, const float* __restrict__ fb2,
                                                          const float* __restrict__ fgamma, const float* __restrict__ frv, int frv_ld,
                                                          uint16_t* __restrict__ qbuf) {
    constexpr int QT = MT * 32, H = C / DH, NTW = C / 128, HD2 = DH / 2, CH = HD2 / 8;
    constexpr int XSTR = C * 2 + 16;  // bytes per XS row (the +16 spreads a 16-lane ds_read_b128 group over all banks)
    constexpr int KSTR = DH * 2 + 16;
    static_assert(C % 128 == 0 && C % DH == 0 && DH % 32 == 0 && MT >= 1 && MT <= 4, "shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ float inv_rev[HD2];
    unsigned char* XS = lds_raw;
    unsigned char* Ks = XS + QT * XSTR;
    const int VS = kc * 2 + 8;
    const int b = blockIdx.y, q0 = blockIdx.x * QT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
    const int nq = qlen ? min(qlen[b], L) : L;
    if (q0 >= nq) return;  // uniform: rows past the utterance's length stay as they are (exact zeros in the padded layout)
    const int nrows = min(QT, nq - q0);
    const int nk = klen ? min(klen[b], Lk) : Lk;
    const int64_t xrow0 = (q_off ? (int64_t)q_off[b] : (int64_t)b * L) + q0;
    const int64_t krow0 = k_off ? (int64_t)k_off[b] : (int64_t)b * Lk;
    if (rope_mode >= 0)
        for (int i = tid; i < HD2; i += 256) inv_rev[i] = __expf(-log_base * (float)(2 * i) / (float)DH) * 0.15915494309189535f;

    XA_STAMP(0);
    const int ncol0 = wave * NTW * 32;  // this wave's output columns in the two projections
    bf16x8_t bfr[XA_PD][NTW];
    float4 keep[MT * 4][C / 128];  // this lane's share of the tile's residual rows (fp32), from here to the final store
    if constexpr (PART != 2) {
    frag_prefetch<C, NTW>(Wq, ncol0, lane, bfr);  // the first four k-steps of Wq travel while the LayerNorm runs

    // ---- 0. (fold +) LayerNorm -> XS --------------------------------------------------------------------------------------
    // One row per HALF wavefront (32 lanes x 3 float4 slots = the 384 channels), two rows of each half in flight: every global
    // load of a pass is issued before the first use, the two LayerNorm reductions are 4 DPP steps + one swizzle (kernels_fold.hpp),
    // and nothing in the loop depends on a runtime "pointer or constant" choice (fold / time-vector presence are compile-time).
    if (!fpart) xa_ln_phase<C, MT, F16, 0, false>(x, xrow0, nrows, XS, XSTR, ln_g, ln_b, eps, nullptr, 0, nullptr, nullptr, nullptr, tid, keep);
    else if (fS == 4 && frv) xa_ln_phase<C, MT, F16, 4, true>(x, xrow0, nrows, XS, XSTR, ln_g, ln_b, eps, fpart, fstride, fb2, fgamma, frv + (size_t)b * frv_ld, tid, keep);
    else if (fS == 4) xa_ln_phase<C, MT, F16, 4, false>(x, xrow0, nrows, XS, XSTR, ln_g, ln_b, eps, fpart, fstride, fb2, fgamma, nullptr, tid, keep);
    else if (fS == 12 && frv) xa_ln_phase<C, MT, F16, 12, true>(x, xrow0, nrows, XS, XSTR, ln_g, ln_b, eps, fpart, fstride, fb2, fgamma, frv + (size_t)b * frv_ld, tid, keep);
    else if (fS == 12) xa_ln_phase<C, MT, F16, 12, false>(x, xrow0, nrows, XS, XSTR, ln_g, ln_b, eps, fpart, fstride, fb2, fgamma, nullptr, tid, keep);
    else if (frv) xa_ln_phase<C, MT, F16, 24, true>(x, xrow0, nrows, XS, XSTR, ln_g, ln_b, eps, fpart, fstride, fb2, fgamma, frv + (size_t)b * frv_ld, tid, keep);
    else xa_ln_phase<C, MT, F16, 24, false>(x, xrow0, nrows, XS, XSTR, ln_g, ln_b, eps, fpart, fstride, fb2, fgamma, nullptr, tid, keep);
    __syncthreads();
    XA_STAMP(1);

    // ---- 1. Q = LN(x) Wq^T + bq -> XS (16-bit), then rotation + softmax scale in place ------------------------------------
    {
        f32x16_t acc[MT][NTW];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;
        rows_times_wt<C, MT, NTW, F16>(XS, XSTR, Wq, ncol0, lane, bfr, acc);
        XA_STAMP(2);
        __syncthreads();  // every wave has read its A fragments: XS may be overwritten
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int col = ncol0 + nt * 32 + lr;
            const float bs = bq ? bq[col] : 0.f;
            // rope_mode < 0 (style blocks): no rotation pass follows and the softmax scale is applied here, before the one rounding to 16
            // bits (the four-launch form rounds q, scales, rounds again); otherwise q is stored unscaled and the rotation pass scales
            const float wmul = rope_mode < 0 ? rsqrtf((float)DH) * 1.44269504088896340736f : 1.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
                    const float v = (acc[mt][nt][i] + bs) * wmul;
                    *reinterpret_cast<uint16_t*>(XS + row * XSTR + col * 2) = (uint16_t)pack_h2<F16>(v, 0.f);
                }
        }
    }
    __syncthreads();
    XA_STAMP(3);
    if (rope_mode >= 0) {  // rotation (+ scale) in place: a thread keeps ONE (head, 16-byte chunk) — its 8 frequencies stay in registers — and walks down the rows
        const float qmul = rsqrtf((float)DH) * 1.44269504088896340736f;
        const float pscale = rope_mode == 1 ? gamma / (float)(nq > 0 ? nq : 1) : 1.f;
        constexpr int TPR = H * CH;               // threads per row (24)
        constexpr int RG = 256 / TPR;             // rows per pass (10)
        const int c = tid % CH, h = (tid / CH) % H, rg = tid / TPR;
        float frq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) frq[e] = inv_rev[c * 8 + e] * pscale;
        if (rg < RG)
            for (int r = rg; r < QT; r += RG) {
                unsigned char* p0 = XS + r * XSTR + (h * DH + c * 8) * 2;
                const u32x4_t w0 = *reinterpret_cast<const u32x4_t*>(p0), w1 = *reinterpret_cast<const u32x4_t*>(p0 + HD2 * 2);
                u32x4_t o0, o1;
                const float pos = (float)(q0 + r);
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    float a0[2], a1[2], y0[2], y1[2];
                    unpack_h2<F16>(w0[e2], a0[0], a0[1]);
                    unpack_h2<F16>(w1[e2], a1[0], a1[1]);
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const float rev = __builtin_amdgcn_fractf(pos * frq[2 * e2 + u]);
                        const float sn = __builtin_amdgcn_sinf(rev), cs = __builtin_amdgcn_cosf(rev);
                        y0[u] = (a0[u] * cs - a1[u] * sn) * qmul;
                        y1[u] = (a1[u] * cs + a0[u] * sn) * qmul;
                    }
                    o0[e2] = pack_h2<F16>(y0[0], y0[1]);
                    o1[e2] = pack_h2<F16>(y1[0], y1[1]);
                }
                *reinterpret_cast<u32x4_t*>(p0) = o0;
                *reinterpret_cast<u32x4_t*>(p0 + HD2 * 2) = o1;
            }
    }

    }  // PART != 2
    if constexpr (PART == 1) {
        __syncthreads();  // the rotated q tile is complete
        // q rows out, whole rows (16 bytes per item, a row's 48 items on neighbouring threads)
        constexpr int CPRW = C / 8;
        for (int idx = tid; idx < nrows * CPRW; idx += 256) {
            const int r = idx / CPRW, c = idx - r * CPRW;
            *reinterpret_cast<u32x4_t*>(qbuf + (xrow0 + r) * C + c * 8) = *reinterpret_cast<const u32x4_t*>(XS + r * XSTR + c * 16);
        }
        if (fpart) {  // the folded rows (fp32) replace x: part 2 adds the block's output to them
            const int l32 = lane & 31, hw = lane >> 5;
#pragma unroll
            for (int k = 0; k < MT * 4; ++k) {
                const int r = 2 * wave + hw + 8 * k;
                if (r < nrows) {
#pragma unroll
                    for (int i = 0; i < C / 128; ++i) reinterpret_cast<float4*>(x + (xrow0 + r) * C)[l32 + 32 * i] = keep[k][i];
                }
            }
        }
        return;
    }
    if constexpr (PART == 2) {
        // the residual rows this lane updates at the end (their loads stay in flight under the whole kernel) and the q tile
        {
            const int l32 = lane & 31, hw = lane >> 5;
#pragma unroll
            for (int k = 0; k < MT * 4; ++k) {
                const int r = 2 * wave + hw + 8 * k;
                const float4* x4 = reinterpret_cast<const float4*>(x + (xrow0 + (r < nrows ? r : 0)) * C);
#pragma unroll
                for (int i = 0; i < C / 128; ++i) keep[k][i] = x4[l32 + 32 * i];
            }
        }
        constexpr int CPRW = C / 8, NQI = (QT * CPRW + 255) / 256;
        u32x4_t qw[NQI];
#pragma unroll
        for (int i = 0; i < NQI; ++i) {
            const int idx = tid + 256 * i, r = idx / CPRW, c = idx - r * CPRW;
            qw[i] = *reinterpret_cast<const u32x4_t*>(qbuf + (xrow0 + (r < nrows ? r : 0)) * C + c * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NQI; ++i) {
            const int idx = tid + 256 * i, r = idx / CPRW, c = idx - r * CPRW;
            if (idx < QT * CPRW) *reinterpret_cast<u32x4_t*>(XS + r * XSTR + c * 16) = r < nrows ? qw[i] : u32x4_t{0u, 0u, 0u, 0u};
        }
    }

    // ---- 2. attention.  SLOTS heads are staged at a time (K rows and V^T per slot); a task = (32-query tile, slot) and wave w
    // takes tasks w, w + 4, ...: with one query tile per workgroup two waves work on two heads at once ---------------------------
    constexpr int SLOTS = MT <= 2 ? 2 : 1;
    static_assert(H % SLOTS == 0, "heads per round");
    const int nkt = kc >> 5;
    const int SLOTB = kc * KSTR + DH * VS;  // bytes per slot
    XA_STAMP(4);
    for (int h0 = 0; h0 < H; h0 += SLOTS) {
        __syncthreads();  // Q complete (first round) / every wave done with the previous round's K and V
        XA_STAMP(5 + 2 * (h0 / SLOTS));
        // K rows as they are, V transposed.  An item = (slot, PAIR of keys, 16-byte chunk of the head): the two keys' values of one
        // dimension make one 4-byte LDS store (half the stores of a key-by-key scatter), and the loads of XA_SU items are all issued
        // before the first LDS store (an item-by-item loop pays one global round trip per item: 9 per round, 8 k of its 8.4 k cycles).
        {
            constexpr int CPK = DH / 8;  // 16-byte chunks per key and head
            const int items = SLOTS * (kc >> 1) * CPK;
            for (int i0 = tid; i0 < items; i0 += 256 * XA_SU) {
                u32x4_t wk[XA_SU][2], wv[XA_SU][2];
#pragma unroll
                for (int u = 0; u < XA_SU; ++u) {
                    const int idx = i0 + u * 256;
                    const int sl = idx / ((kc >> 1) * CPK), rem = idx - sl * ((kc >> 1) * CPK);
                    const int kl = (rem / CPK) * 2, c = rem % CPK;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        wk[u][e] = u32x4_t{0u, 0u, 0u, 0u}; wv[u][e] = wk[u][e];
                        if (idx < items && kl + e < nk) {
                            wk[u][e] = *reinterpret_cast<const u32x4_t*>(kp + (krow0 + kl + e) * ldk + (h0 + sl) * DH + c * 8);
                            wv[u][e] = *reinterpret_cast<const u32x4_t*>(vp + (krow0 + kl + e) * ldk + (h0 + sl) * DH + c * 8);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < XA_SU; ++u) {
                    const int idx = i0 + u * 256;
                    if (idx >= items) break;
                    const int sl = idx / ((kc >> 1) * CPK), rem = idx - sl * ((kc >> 1) * CPK);
                    const int kl = (rem / CPK) * 2, c = rem % CPK;
                    unsigned char* Kb = Ks + sl * SLOTB;
                    unsigned char* Vb = Kb + kc * KSTR;
                    *reinterpret_cast<u32x4_t*>(Kb + kl * KSTR + c * 16) = wk[u][0];
                    *reinterpret_cast<u32x4_t*>(Kb + (kl + 1) * KSTR + c * 16) = wk[u][1];
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) {
                        const unsigned a0 = wv[u][0][e2], a1 = wv[u][1][e2];
                        *reinterpret_cast<unsigned*>(Vb + (c * 8 + 2 * e2) * VS + kl * 2) = (a0 & 0xFFFFu) | (a1 << 16);
                        *reinterpret_cast<unsigned*>(Vb + (c * 8 + 2 * e2 + 1) * VS + kl * 2) = (a0 >> 16) | (a1 & 0xFFFF0000u);
                    }
                }
            }
        }
        if (h0 + SLOTS >= H) frag_prefetch<C, NTW>(Wo, ncol0, lane, bfr);  // the output projection's first k-steps travel during the last round
        __syncthreads();
        XA_STAMP(6 + 2 * (h0 / SLOTS));
#pragma unroll 1
        for (int task = wave; task < MT * SLOTS; task += 4) {
            const int qbase = (task % MT) * 32, sl = task / MT, h = h0 + sl;
            if (qbase >= nrows) continue;  // wave-uniform
            const unsigned char* Kb = Ks + sl * SLOTB;
            const unsigned char* Vb = Kb + kc * KSTR;
            bf16x8_t bqf[DH / 16];
#pragma unroll
            for (int ks = 0; ks < DH / 16; ++ks)
                bqf[ks] = *reinterpret_cast<const bf16x8_t*>(XS + (qbase + lr) * XSTR + h * DH * 2 + (ks * 2 + lh) * 16);
            // all scores of the chunk (<= 4 key tiles of 32) stay in registers: one QK^T pass, the row maximum, the exponentials and
            // their row sum, then P is scaled by 1 / sum ON THE LANE (a lane is a query in this layout) before it becomes the A operand
            // of P V — the output needs no per-row normalisation (16 ds_bpermute per task in the two-pass form)
            f32x16_t sc[4];
            float m = -1e30f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                if (kt < nkt) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) sc[kt][i] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < DH / 16; ++ks) {
                        const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(Kb + (kt * 32 + lr) * KSTR + (ks * 2 + lh) * 16);
                        sc[kt] = mfma_h<F16>(a, bqf[ks], sc[kt]);
                    }
                }
            }
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
                if (kt < nkt) {
                    asm volatile("s_nop 7" : "+v"(sc[kt]));  // (cheap) keeps the reads below behind the MFMAs whatever the block layout
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
                        sc[kt][i] = key < nk ? sc[kt][i] : -1e30f;
                        m = fmaxf(m, sc[kt][i]);
                    }
                }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float lsum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
                if (kt < nkt) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float pv = sc[kt][i] > -1e29f ? exp2f(sc[kt][i] - m) : 0.f;
                        sc[kt][i] = pv;
                        lsum += pv;
                    }
                }
            lsum += __shfl_xor(lsum, 32, 64);
            const float inv = (nk > 0 && lsum > 0.f) ? 1.0f / lsum : 0.f;
            f32x16_t oacc[DH / 32];
#pragma unroll
            for (int nd = 0; nd < DH / 32; ++nd)
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[nd][i] = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
                if (kt < nkt) {
#pragma unroll
                    for (int sidx = 0; sidx < 2; ++sidx) {
                        u32x4_t pw;
                        pw[0] = pack_h2<F16>(sc[kt][8 * sidx + 0] * inv, sc[kt][8 * sidx + 1] * inv);
                        pw[1] = pack_h2<F16>(sc[kt][8 * sidx + 2] * inv, sc[kt][8 * sidx + 3] * inv);
                        pw[2] = pack_h2<F16>(sc[kt][8 * sidx + 4] * inv, sc[kt][8 * sidx + 5] * inv);
                        pw[3] = pack_h2<F16>(sc[kt][8 * sidx + 6] * inv, sc[kt][8 * sidx + 7] * inv);
                        const bf16x8_t ap = __builtin_bit_cast(bf16x8_t, pw);
#pragma unroll
                        for (int nd = 0; nd < DH / 32; ++nd) {
                            const unsigned char* base = Vb + (nd * 32 + lr) * VS + (kt * 32 + 16 * sidx + 4 * lh) * 2;
                            const uint2 lo = *reinterpret_cast<const uint2*>(base);
                            const uint2 hi = *reinterpret_cast<const uint2*>(base + 16);
                            u32x4_t vw;
                            vw[0] = lo.x; vw[1] = lo.y; vw[2] = hi.x; vw[3] = hi.y;
                            oacc[nd] = mfma_h<F16>(ap, __builtin_bit_cast(bf16x8_t, vw), oacc[nd]);
                        }
                    }
                }
            // wait states between the last MFMA and the first read of its accumulators (see kernels_attn.hip)
            if constexpr (DH == 32) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(oacc[0]));
            else if constexpr (DH == 64) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(oacc[0]), "+a"(oacc[1]));
            else asm volatile("s_nop 15\n\ts_nop 7" : "+a"(oacc[0]), "+a"(oacc[1]), "+a"(oacc[2]));
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int qrow = (i & 3) + 8 * (i >> 2) + 4 * lh;
#pragma unroll
                for (int nd = 0; nd < DH / 32; ++nd)  // O_h over Q_h: only this task reads or writes these rows of head h
                    *reinterpret_cast<uint16_t*>(XS + (qbase + qrow) * XSTR + (h * DH + nd * 32 + lr) * 2) = (uint16_t)pack_h2<F16>(oacc[nd][i], 0.f);
            }
        }
    }
    __syncthreads();
    XA_STAMP(13);

    // ---- 3. x += O Wo^T + bo ---------------------------------------------------------------------------------------------
    {
        f32x16_t acc[MT][NTW];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;
        rows_times_wt<C, MT, NTW, F16>(XS, XSTR, Wo, ncol0, lane, bfr, acc);
        XA_STAMP(14);
        // residual update through LDS: the accumulators (+ bias) of ALL the tile's rows go to one fp32 image [QT][C + 4] over XS and
        // the K / V slots (both dead once every wave has left the MFMA loop), then every lane adds its rows' pieces of it to the fp32
        // rows it has kept since phase 0 and stores x
        float* YS = reinterpret_cast<float*>(lds_raw);
        constexpr int YSTR = C + 4;  // floats per image row (+4: rows start on different banks)
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int col = ncol0 + nt * 32 + lr;
                const float bs = bo ? bo[col] : 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) YS[(mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh) * YSTR + col] = acc[mt][nt][i] + bs;
            }
        __syncthreads();
        {
            const int l32 = lane & 31, hw = lane >> 5;
#pragma unroll
            for (int k = 0; k < MT * 4; ++k) {
                const int r = 2 * wave + hw + 8 * k;  // the rows this lane kept in phase 0
                if (r < nrows) {
#pragma unroll
                    for (int i = 0; i < C / 128; ++i) {
                        const float4 yv = *reinterpret_cast<const float4*>(YS + r * YSTR + (l32 + 32 * i) * 4), xo = keep[k][i];
                        reinterpret_cast<float4*>(x + (xrow0 + r) * C)[l32 + 32 * i] = make_float4(xo.x + yv.x, xo.y + yv.y, xo.z + yv.z, xo.w + yv.w);
                    }
                }
            }
        }
    }
    XA_STAMP(15);
}

template <int C, int DH, int MT, bool F16, int PART>
void launch_one(hipStream_t s, float* x, const float* ln_g, const float* ln_b, float eps, const void* Wq, const float* bq, const void* kp,
                const void* vp, int ldk, const void* Wo, const float* bo, int B, int L, int Lk, int kc, const int* qlen, const int* klen,
                const int* q_off, const int* k_off, int rope_mode, float log_base, float gamma, const FoldArgs* fold, void* qbuf) {
    constexpr int SLOTS = MT <= 2 ? 2 : 1;  // heads staged at a time (see the kernel)
    const size_t kv = SLOTS * ((size_t)kc * (DH * 2 + 16) + (size_t)DH * (kc * 2 + 8)), ys = (size_t)MT * 32 * (C + 4) * 4;  // K/V slots; the residual image (over everything)
    const size_t lds = PART == 1 ? (size_t)MT * 32 * (C * 2 + 16) : std::max((size_t)MT * 32 * (C * 2 + 16) + kv, ys);
    static PerDeviceOnce attr_once;
    if (attr_once.need())
        stn_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_fused_kernel<C, DH, MT, F16, PART>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024),
                      "hipFuncSetAttribute(xattn_fused)");
    const dim3 grid((L + MT * 32 - 1) / (MT * 32), B);
    STN_KLAUNCH((xattn_fused_kernel<C, DH, MT, F16, PART>), grid, dim3(256), lds, s, x, ln_g, ln_b, eps, static_cast<const uint16_t*>(Wq), bq,
                static_cast<const uint16_t*>(kp), static_cast<const uint16_t*>(vp), ldk, static_cast<const uint16_t*>(Wo), bo, L, Lk, kc, qlen,
                klen, q_off, k_off, rope_mode, log_base, gamma, fold ? static_cast<const uint16_t*>(fold->part) : nullptr, fold ? fold->S : 0,
                fold ? fold->part_stride : 0, fold ? fold->b2 : nullptr, fold ? fold->gamma : nullptr, fold ? fold->rowvec : nullptr, fold ? fold->rv_ld : 0,
                static_cast<uint16_t*>(qbuf));
}

}  // namespace

bool xattn_fused_supported(int dtype, int C, int H, int Lk, int ldk) {
    return is_half(dtype) && C == 384 && H == 4 && Lk >= 1 && Lk <= 128 && ldk % 8 == 0;
}

void launch_xattn_fused(hipStream_t s, int dtype, float* x, const float* ln_g, const float* ln_b, float eps, const void* Wq, const float* bq,
                        const void* kp, const void* vp, int ldk, const void* Wo, const float* bo, int B, int L, int C, int H, int Lk,
                        const int* qlen, const int* klen, const int* q_off, const int* k_off, int rope_mode, float rope_base, float rope_gamma,
                        const FoldArgs* fold, int part, void* qbuf) {
    if (B == 0 || L == 0) return;
    if (part < 0 || part > 2 || (part && (!qbuf || (reinterpret_cast<uintptr_t>(qbuf) & 15)))) throw std::invalid_argument("launch_xattn_fused: part 1 / 2 need a 16-byte aligned q buffer of rows x C");
    if (fold && (!fold->part || (fold->S != 4 && fold->S != 12 && fold->S != 24) || !fold->b2 || !fold->gamma || (fold->rowvec && fold->rv_ld % 4)))
        throw std::invalid_argument("launch_xattn_fused: the pending fold needs 16-bit partial sums of 4, 12 or 24 splits, b2 and gamma");
    if (!xattn_fused_supported(dtype, C, H, Lk, ldk) || (q_off && !qlen) || (k_off && !klen) || (reinterpret_cast<uintptr_t>(kp) & 15) ||
        (reinterpret_cast<uintptr_t>(vp) & 15) || (reinterpret_cast<uintptr_t>(Wq) & 15) || (reinterpret_cast<uintptr_t>(Wo) & 15)) { throw std::invalid_argument("launch_xattn_fused: unsupported shape or alignment (callers check xattn_fused_supported)"); }
    const int kc = (Lk + 31) & ~31;
    const float lb = logf(rope_base);
    // rows per workgroup: the smallest tile that still leaves about two workgroups per CU, so short utterances spread over the chip
    const long t32 = (long)B * ((L + 31) / 32), t64 = (long)B * ((L + 63) / 64);
    int mt = t32 <= 256 ? 1 : (t64 <= 768 ? 2 : 4);  // measured at B = 128, L = 78: 64-row tiles 15.5 ms per batch, 32- and 128-row tiles 16.2
    if (const char* f = getenv("STN_XATTN_MT")) mt = atoi(f) == 4 ? 4 : (atoi(f) == 2 ? 2 : 1);  // experiments
#define STN_XA(MT_, F16_, P_) launch_one<384, 96, MT_, F16_, P_>(s, x, ln_g, ln_b, eps, Wq, bq, kp, vp, ldk, Wo, bo, B, L, Lk, kc, qlen, klen, q_off, k_off, rope_mode, lb, rope_gamma, fold, qbuf)
    if (part == 1) {  // rows only: 32-row tiles while they leave at most ~4 workgroups per CU, 64 beyond
        if (t32 <= 1024) { if (dtype == F16) STN_XA(1, true, 1); else STN_XA(1, false, 1); }
        else { if (dtype == F16) STN_XA(2, true, 1); else STN_XA(2, false, 1); }
        return;
    }
    if (part == 2) {
        if (dtype == F16) { if (mt == 1) STN_XA(1, true, 2); else if (mt == 2) STN_XA(2, true, 2); else STN_XA(4, true, 2); }
        else { if (mt == 1) STN_XA(1, false, 2); else if (mt == 2) STN_XA(2, false, 2); else STN_XA(4, false, 2); }
        return;
    }
    if (dtype == F16) { if (mt == 1) STN_XA(1, true, 0); else if (mt == 2) STN_XA(2, true, 0); else STN_XA(4, true, 0); }
    else { if (mt == 1) STN_XA(1, false, 0); else if (mt == 2) STN_XA(2, false, 0); else STN_XA(4, false, 0); }
#undef STN_XA
}

}  // namespace stn
