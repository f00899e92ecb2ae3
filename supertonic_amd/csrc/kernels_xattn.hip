// kernels_xattn.hip — one launch per cross-attention block of the vector estimator (gfx950, wave64, bf16 / f16 modes):
//
//     x <- x + Wo . attention(q = Wq . LN(x) + bq,  K, V) + bo          for the rows one utterance owns
//
// The unfused block is four launches (LayerNorm, q GEMM, attention, output GEMM + residual) of 6-15 us each around ~2 us of
// matrix work; 40 of them run per 128-utterance batch (4 blocks x text/style x 5 Euler steps), ~11 % of the batch.  Here one
// workgroup (4 waves) owns a tile of QT = 32*MT query rows of one utterance and keeps everything between the two global
// accesses of x in LDS / registers:
//   0. LayerNorm of the tile's rows (one wave per row, the arithmetic of dwconv_ln_kernel<.., false>) -> XS [QT][C] (16-bit)
//   1. Q = XS . Wq^T: wave w owns output columns [w*C/4, (w+1)*C/4); A fragments from XS (ds_read_b128), B fragments straight
//      from global memory in the MFMA operand layout (lane = output column, 8 consecutive k = 16 B), prefetched 4 k-steps
//      ahead in registers; Q + bias is written back over XS as 16-bit, then rotated (LARoPE) and scaled in place — the same
//      two roundings the unfused path makes (GEMM output, attention staging)
//   2. per head: K / V^T of the context staged in LDS (keys arrive already rotated), QK^T / softmax / PV exactly as
//      attn_mfma_kernel's single-chunk path (keys on the accumulator rows, two-pass softmax); O_h overwrites Q_h in XS
//   3. Y = XS . Wo^T like step 1, and x += Y + bo straight from the accumulators (4-byte accesses, 128 B per half-wave)
// Contexts longer than one 128-key chunk, other widths and fp32 engines take the unfused path (Engine::ve_step_dev).
//
// STATUS (round 1): correct (tests/test_gpu_xattn.py: equal to the four-launch form far inside one 16-bit rounding, both layouts,
// bf16 and f16) but SLOWER at batch 128: 15.5 ms per batch with 64-row tiles (16.2 with 32- or 128-row tiles) against 14.9 ms for
// the four launches.  In-kernel stamps (-DXA_STAMPS) of a 32-row workgroup after the second pass over the kernel (interleaved
// LayerNorm reductions, scores kept in registers with P normalised on the lane, residual update through an LDS image): LayerNorm
// 17 k cycles, q projection 8-15 k, q write-back + rotation 8 k, attention 2 x (staging 9 k + compute 8 k), output projection 18 k,
// residual 11 k = 105 k cycles.  Two things bound it: every 32-row tile streams both weight matrices (576 KB) through its CU's L1
// (the four-launch GEMMs read them once per 128 rows), and with 4-8 waves per CU each phase runs at the latency of one wave's
// dependent chain.  The engine therefore keeps the four-launch form by default (stn_set_fused_xattn / STN_XATTN=1 select this
// kernel); see DESIGN.md section 9 for what would have to change.
#include "kernels.hpp"

#include <stdio.h>
#include <stdlib.h>

namespace stn {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    const unsigned ra = (ua + 0x7FFFu + ((ua >> 16) & 1u)) >> 16, rb = (ub + 0x7FFFu + ((ub >> 16) & 1u)) >> 16;
    return ra | (rb << 16);
}
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
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// acc[mt][nt] += XS[32*mt .. +32][:] . W[n0 + 32*nt .. +32][:]^T over K = C.  The weight arrives in FRAGMENT ORDER
// (launch_repack_frag): the 64 x 16 bytes one MFMA B operand needs — lane (lr, lh) of n-tile T, k-step ks holds
// W[32*T + lr][16*ks + 8*lh .. +8] — are one contiguous KiB at ((T * C/16 + ks) * 64 + lane) * 16 bytes, so every load is a
// fully coalesced wave-wide KiB and consecutive k-steps are consecutive KiBs.  (Read straight from the row-major matrix the
// same loads touch 32 lines for 32 bytes each: measured 85 us per block instead of the four-launch form's 43.)
// Four k-steps stay in flight in registers; `bfr` arrives pre-loaded with k-steps 0..3 (frag_prefetch), so that a caller can
// issue them ahead of unrelated work.
template <int C, int NTW>
__device__ __forceinline__ void frag_prefetch(const uint16_t* __restrict__ Wf, int n0, int lane, bf16x8_t (&bfr)[4][NTW]) {
    constexpr int NKS = C / 16;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
            bfr[p][nt] = *reinterpret_cast<const bf16x8_t*>(Wf + ((size_t)((n0 / 32 + nt) * NKS + p) * 64 + lane) * 8);
}
template <int C, int MT, int NTW, bool F16>
__device__ __forceinline__ void rows_times_wt(const unsigned char* __restrict__ XS, int XSTR, const uint16_t* __restrict__ Wf, int n0,
                                              int lane, bf16x8_t (&bfr)[4][NTW], f32x16_t (&acc)[MT][NTW]) {
    constexpr int NKS = C / 16, PD = 4;
    static_assert(NKS % PD == 0, "K must be a multiple of 64");
    const int lr = lane & 31, lh = lane >> 5;
    for (int ks0 = 0; ks0 < NKS; ks0 += PD) {
#pragma unroll
        for (int p = 0; p < PD; ++p) {
            const int ks = ks0 + p;
            bf16x8_t a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const bf16x8_t*>(XS + (mt * 32 + lr) * XSTR + (ks * 2 + lh) * 16);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = mfma_h<F16>(a[mt], bfr[p][nt], acc[mt][nt]);
            if (ks + PD < NKS) {
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    bfr[p][nt] = *reinterpret_cast<const bf16x8_t*>(Wf + ((size_t)((n0 / 32 + nt) * NKS + ks + PD) * 64 + lane) * 8);
            }
        }
    }
}

// [N][K] row-major 16-bit matrix -> fragment order (see rows_times_wt); N % 32 == 0, K % 16 == 0
__global__ void repack_frag_kernel(const uint16_t* __restrict__ W, int N, int K, uint16_t* __restrict__ Wf) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte piece each
    const int NKS = K / 16;
    if (idx >= (int64_t)(N / 32) * NKS * 64) return;
    const int lane = (int)(idx & 63);
    const int64_t blk = idx >> 6;
    const int ks = (int)(blk % NKS), T = (int)(blk / NKS);
    const int lr = lane & 31, lh = lane >> 5;
    reinterpret_cast<u32x4_t*>(Wf)[idx] = *reinterpret_cast<const u32x4_t*>(W + (size_t)(T * 32 + lr) * K + ks * 16 + lh * 8);
}

#ifdef XA_STAMPS
__device__ unsigned long long g_xa_ts[16];
extern "C" void stn_dbg_xa(unsigned long long* out) { (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xa_ts), sizeof(unsigned long long) * 16); }
#define XA_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 5 && threadIdx.x == 0) g_xa_ts[i] = __builtin_readcyclecounter(); } while (0)
#else
#define XA_STAMP(i) do { } while (0)
#endif
template <int C, int DH, int MT, bool F16>
__global__ __launch_bounds__(256) void xattn_fused_kernel(float* __restrict__ x, const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                          float eps, const uint16_t* __restrict__ Wq, const float* __restrict__ bq,
                                                          const uint16_t* __restrict__ kp, const uint16_t* __restrict__ vp, int ldk,
                                                          const uint16_t* __restrict__ Wo, const float* __restrict__ bo, int L, int Lk,
                                                          int kc /* keys in the LDS chunk: multiple of 32, <= 128, >= Lk */,
                                                          const int* __restrict__ qlen, const int* __restrict__ klen,
                                                          const int* __restrict__ q_off, const int* __restrict__ k_off, int rope_mode,
                                                          float log_base, float gamma) {
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
    bf16x8_t bfr[4][NTW];
    frag_prefetch<C, NTW>(Wq, ncol0, lane, bfr);  // the first four k-steps of Wq travel while the LayerNorm runs

    // ---- 0. LayerNorm -> XS --------------------------------------------------------------------------------------------
    // a wave owns rows wave, wave + 4, ...; the rows of one pass (8 per wave) are all requested before the first is reduced, so
    // a pass pays one global round trip instead of eight
    {
        constexpr int C4 = C / 4, NI = (C4 + 63) / 64, RP = 8;
        const float4* g4 = reinterpret_cast<const float4*>(ln_g);
        const float4* b4 = reinterpret_cast<const float4*>(ln_b);
#pragma unroll 1
        for (int pass = 0; pass < MT; ++pass) {
            float4 h[RP][NI];
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                const int r = wave + 4 * (pass * RP + j);
                const float4* x4 = reinterpret_cast<const float4*>(x + (xrow0 + r) * C);
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    h[j][i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (r < nrows && lane + 64 * i < C4) h[j][i] = x4[lane + 64 * i];
                }
            }
            // the RP rows' two reductions run interleaved (RP independent butterflies per step instead of RP dependent chains)
            float sm[RP], vr[RP];
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                sm[j] = 0.f;
#pragma unroll
                for (int i = 0; i < NI; ++i) sm[j] += (h[j][i].x + h[j][i].y) + (h[j][i].z + h[j][i].w);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
#pragma unroll
                for (int j = 0; j < RP; ++j) sm[j] += __shfl_xor(sm[j], o, 64);
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                sm[j] = sm[j] / (float)C;  // mean
                vr[j] = 0.f;
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    if (lane + 64 * i < C4) {
                        const float dx = h[j][i].x - sm[j], dy = h[j][i].y - sm[j], dz = h[j][i].z - sm[j], dw = h[j][i].w - sm[j];
                        vr[j] += (dx * dx + dy * dy) + (dz * dz + dw * dw);
                    }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
#pragma unroll
                for (int j = 0; j < RP; ++j) vr[j] += __shfl_xor(vr[j], o, 64);
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                const int r = wave + 4 * (pass * RP + j);
                const bool live = r < nrows;  // rows of the tile beyond the utterance: zeros (their MFMA rows are never stored)
                const float mean = sm[j], rstd = rsqrtf(vr[j] / (float)C + eps);
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int c4 = lane + 64 * i;
                    if (c4 < C4) {
                        const float4 gg = g4[c4], bb = b4[c4];
                        uint2 u = make_uint2(0u, 0u);
                        if (live) {
                            u.x = pack_h2<F16>((h[j][i].x - mean) * rstd * gg.x + bb.x, (h[j][i].y - mean) * rstd * gg.y + bb.y);
                            u.y = pack_h2<F16>((h[j][i].z - mean) * rstd * gg.z + bb.z, (h[j][i].w - mean) * rstd * gg.w + bb.w);
                        }
                        *reinterpret_cast<uint2*>(XS + r * XSTR + c4 * 8) = u;
                    }
                }
            }
        }
    }
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
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
                    *reinterpret_cast<uint16_t*>(XS + row * XSTR + col * 2) = (uint16_t)pack_h2<F16>(acc[mt][nt][i] + bs, 0.f);
                }
        }
    }
    __syncthreads();
    XA_STAMP(3);
    {
        const float qmul = rsqrtf((float)DH) * 1.44269504088896340736f;
        const bool rot = rope_mode >= 0;
        const float pscale = rope_mode == 1 ? gamma / (float)(nq > 0 ? nq : 1) : 1.f;
        for (int idx = tid; idx < QT * H * CH; idx += 256) {
            const int r = idx / (H * CH), rem = idx - r * (H * CH), h = rem / CH, c = rem - h * CH;
            unsigned char* p0 = XS + r * XSTR + (h * DH + c * 8) * 2;
            const u32x4_t w0 = *reinterpret_cast<const u32x4_t*>(p0), w1 = *reinterpret_cast<const u32x4_t*>(p0 + HD2 * 2);
            u32x4_t o0, o1;
            const float pp = (float)(q0 + r) * pscale;
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                float a0[2], a1[2], y0[2], y1[2];
                unpack_h2<F16>(w0[e2], a0[0], a0[1]);
                unpack_h2<F16>(w1[e2], a1[0], a1[1]);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float cs = 1.f, sn = 0.f;
                    if (rot) {
                        const float rev = __builtin_amdgcn_fractf(pp * inv_rev[c * 8 + 2 * e2 + u]);
                        sn = __builtin_amdgcn_sinf(rev);
                        cs = __builtin_amdgcn_cosf(rev);
                    }
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
        for (int idx = tid; idx < SLOTS * kc * (DH / 8); idx += 256) {
            const int sl = idx / (kc * (DH / 8)), rem = idx - sl * (kc * (DH / 8));
            const int kl = rem / (DH / 8), c = rem - kl * (DH / 8);
            unsigned char* Kb = Ks + sl * SLOTB;
            unsigned char* Vb = Kb + kc * KSTR;
            u32x4_t wk = {0u, 0u, 0u, 0u}, wv = wk;
            if (kl < nk) {
                wk = *reinterpret_cast<const u32x4_t*>(kp + (krow0 + kl) * ldk + (h0 + sl) * DH + c * 8);
                wv = *reinterpret_cast<const u32x4_t*>(vp + (krow0 + kl) * ldk + (h0 + sl) * DH + c * 8);
            }
            *reinterpret_cast<u32x4_t*>(Kb + kl * KSTR + c * 16) = wk;
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                *reinterpret_cast<uint16_t*>(Vb + (c * 8 + 2 * e2) * VS + kl * 2) = (uint16_t)(wv[e2] & 0xFFFFu);
                *reinterpret_cast<uint16_t*>(Vb + (c * 8 + 2 * e2 + 1) * VS + kl * 2) = (uint16_t)(wv[e2] >> 16);
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
        // residual update through LDS: the accumulators (+ bias) go to an fp32 image [QT][C] over the K / V slots (dead by now), then
        // every thread adds 16-byte pieces of it to x — 12 independent float4 read-modify-writes per thread and 32-row tile instead of
        // 48 dependent 4-byte ones per lane
        float* YS = reinterpret_cast<float*>(Ks);  // [32][C + 4] fp32: one 32-row tile at a time (the launcher sizes the region)
        constexpr int YSTR = C + 4;  // floats per image row (+4: rows start on different banks)
        constexpr int C4 = C / 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (mt) __syncthreads();  // the previous tile's image has been consumed
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int col = ncol0 + nt * 32 + lr;
                const float bs = bo ? bo[col] : 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) YS[((i & 3) + 8 * (i >> 2) + 4 * lh) * YSTR + col] = acc[mt][nt][i] + bs;
            }
            __syncthreads();
            for (int idx0 = 0; idx0 < 32 * C4; idx0 += 256 * 4) {
                float4 xv[4];
                int rr[4], cc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = idx0 + u * 256 + tid;
                    rr[u] = idx / C4; cc[u] = idx - rr[u] * C4;
                    xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (idx < 32 * C4 && mt * 32 + rr[u] < nrows) xv[u] = *reinterpret_cast<const float4*>(x + (xrow0 + mt * 32 + rr[u]) * C + cc[u] * 4);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = idx0 + u * 256 + tid;
                    if (idx < 32 * C4 && mt * 32 + rr[u] < nrows) {
                        const float4 yv = *reinterpret_cast<const float4*>(YS + rr[u] * YSTR + cc[u] * 4);
                        *reinterpret_cast<float4*>(x + (xrow0 + mt * 32 + rr[u]) * C + cc[u] * 4) = make_float4(xv[u].x + yv.x, xv[u].y + yv.y, xv[u].z + yv.z, xv[u].w + yv.w);
                    }
                }
            }
        }
    }
    XA_STAMP(15);
}

template <int C, int DH, int MT, bool F16>
void launch_one(hipStream_t s, float* x, const float* ln_g, const float* ln_b, float eps, const void* Wq, const float* bq, const void* kp,
                const void* vp, int ldk, const void* Wo, const float* bo, int B, int L, int Lk, int kc, const int* qlen, const int* klen,
                const int* q_off, const int* k_off, int rope_mode, float log_base, float gamma) {
    constexpr int SLOTS = MT <= 2 ? 2 : 1;  // heads staged at a time (see the kernel)
    const size_t kv = SLOTS * ((size_t)kc * (DH * 2 + 16) + (size_t)DH * (kc * 2 + 8)), ys = (size_t)32 * (C + 4) * 4;  // K/V slots, later the residual image
    const size_t lds = (size_t)MT * 32 * (C * 2 + 16) + (kv > ys ? kv : ys);
    static PerDeviceOnce attr_once;
    if (attr_once.need())
        stn_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_fused_kernel<C, DH, MT, F16>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024),
                      "hipFuncSetAttribute(xattn_fused)");
    const dim3 grid((L + MT * 32 - 1) / (MT * 32), B);
    STN_KLAUNCH((xattn_fused_kernel<C, DH, MT, F16>), grid, dim3(256), lds, s, x, ln_g, ln_b, eps, static_cast<const uint16_t*>(Wq), bq,
                static_cast<const uint16_t*>(kp), static_cast<const uint16_t*>(vp), ldk, static_cast<const uint16_t*>(Wo), bo, L, Lk, kc, qlen,
                klen, q_off, k_off, rope_mode, log_base, gamma);
}

}  // namespace

void launch_repack_frag(hipStream_t s, const void* W, int N, int K, void* Wf) {
    if (N % 32 || K % 16) { throw std::invalid_argument("launch_repack_frag: N % 32 and K % 16 must be 0"); }
    const int64_t n = (int64_t)(N / 32) * (K / 16) * 64;
    STN_KLAUNCH(repack_frag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, static_cast<const uint16_t*>(W), N, K, static_cast<uint16_t*>(Wf));
}

bool xattn_fused_supported(int dtype, int C, int H, int Lk, int ldk) {
    return is_half(dtype) && C == 384 && H == 4 && Lk >= 1 && Lk <= 128 && ldk % 8 == 0;
}

void launch_xattn_fused(hipStream_t s, int dtype, float* x, const float* ln_g, const float* ln_b, float eps, const void* Wq, const float* bq,
                        const void* kp, const void* vp, int ldk, const void* Wo, const float* bo, int B, int L, int C, int H, int Lk,
                        const int* qlen, const int* klen, const int* q_off, const int* k_off, int rope_mode, float rope_base, float rope_gamma) {
    if (B == 0 || L == 0) return;
    if (!xattn_fused_supported(dtype, C, H, Lk, ldk) || (q_off && !qlen) || (k_off && !klen) || (reinterpret_cast<uintptr_t>(kp) & 15) ||
        (reinterpret_cast<uintptr_t>(vp) & 15) || (reinterpret_cast<uintptr_t>(Wq) & 15) || (reinterpret_cast<uintptr_t>(Wo) & 15)) { throw std::invalid_argument("launch_xattn_fused: unsupported shape or alignment (callers check xattn_fused_supported)"); }
    const int kc = (Lk + 31) & ~31;
    const float lb = logf(rope_base);
    // rows per workgroup: the smallest tile that still leaves about two workgroups per CU, so short utterances spread over the chip
    const long t32 = (long)B * ((L + 31) / 32), t64 = (long)B * ((L + 63) / 64);
    int mt = t32 <= 256 ? 1 : (t64 <= 768 ? 2 : 4);  // measured at B = 128, L = 78: 64-row tiles 15.5 ms per batch, 32- and 128-row tiles 16.2
    if (const char* f = getenv("STN_XATTN_MT")) mt = atoi(f) == 4 ? 4 : (atoi(f) == 2 ? 2 : 1);  // experiments
#define STN_XA(MT_, F16_) launch_one<384, 96, MT_, F16_>(s, x, ln_g, ln_b, eps, Wq, bq, kp, vp, ldk, Wo, bo, B, L, Lk, kc, qlen, klen, q_off, k_off, rope_mode, lb, rope_gamma)
    if (dtype == F16) { if (mt == 1) STN_XA(1, true); else if (mt == 2) STN_XA(2, true); else STN_XA(4, true); }
    else { if (mt == 1) STN_XA(1, false); else if (mt == 2) STN_XA(2, false); else STN_XA(4, false); }
#undef STN_XA
}

}  // namespace stn
