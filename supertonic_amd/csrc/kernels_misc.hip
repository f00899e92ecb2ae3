// kernels_misc.hip — the HBM-bound kernels around the GEMMs (gfx950, wave64).
//
//   dwconv_ln     depthwise dilated conv along time fused with LayerNorm over channels: one wavefront per
//                 frame, 16-B (float4) coalesced channel loads, mean/variance by wavefront shuffles; the
//                 k taps re-read neighbouring frames through L1/L2, HBM sees each frame once.
//   layernorm     same reduction without the conv.
//   vocoder_in    latent un-compress + the ld->C input conv of the vocoder, frame window staged in LDS.
//   + gather / mask / transpose / noise / pcm helpers.
// All row-major [rows][channels]; see kernels.hpp for the contracts.
#include "kernels.hpp"
#include "dev_env.hpp"
#include "kernels_fold.hpp"

#include <hip/hip_bf16.h>
#include <type_traits>
#include <stdio.h>
#include <stdlib.h>

namespace stn {

__device__ __forceinline__ uint16_t f2bf_(float f) {
    __hip_bfloat16 h = __float2bfloat16(f);
    return *reinterpret_cast<uint16_t*>(&h);
}
__device__ __forceinline__ float bf2f_(uint16_t v) { return __uint_as_float(((unsigned)v) << 16); }

__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
__device__ __forceinline__ void store4(uint16_t* p, float a, float b, float c, float d) {
    uint2 u;
    u.x = (unsigned)f2bf_(a) | ((unsigned)f2bf_(b) << 16);
    u.y = (unsigned)f2bf_(c) | ((unsigned)f2bf_(d) << 16);
    *reinterpret_cast<uint2*>(p) = u;
}
__device__ __forceinline__ void store4(f16_t* p, float a, float b, float c, float d) {
    typedef _Float16 h4_ __attribute__((ext_vector_type(4)));
    h4_ h = {(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)d};  // v_cvt_f16_f32: round to nearest even
    *reinterpret_cast<h4_*>(p) = h;
}
__device__ __forceinline__ void store1(float* p, float a) { *p = a; }
__device__ __forceinline__ void store1(uint16_t* p, float a) { *p = f2bf_(a); }
__device__ __forceinline__ void store1(f16_t* p, float a) { *p = (f16_t)a; }
__device__ __forceinline__ float load1(const float* p) { return *p; }
__device__ __forceinline__ float load1(const uint16_t* p) { return bf2f_(*p); }
__device__ __forceinline__ float load1(const f16_t* p) { return (float)*p; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------
// depthwise conv + LayerNorm  (CONV=false: LayerNorm only).  One wavefront per frame.
// ---------------------------------------------------------------------------------------------
static constexpr int LN_NI = 4;  // float4 slots per lane -> C <= 4 * 64 * LN_NI = 1024

template <typename OutT, bool CONV>
__global__ __launch_bounds__(256) void dwconv_ln_kernel(const float* __restrict__ x, int64_t M, int L, int C,
                                                        const float* __restrict__ w_t, const float* __restrict__ bias,
                                                        int k, int dil, const float* __restrict__ g,
                                                        const float* __restrict__ bt, float eps, OutT* __restrict__ y,
                                                        const int* __restrict__ seqlen) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;  // wave-uniform
    const int Lv = (CONV && seqlen) ? seqlen[row / L] : L;       // valid frames of this row's sequence
    const float live = (CONV && (int)(row % L) >= Lv) ? 0.f : 1.f;  // rows in the padding are written as zeros
    const int C4 = C >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float4 h[LN_NI];
    if (CONV) {
        const int t = (int)(row % L);
        const int64_t base = row - t;
        const int half = (k - 1) >> 1;
        const float4* w4 = reinterpret_cast<const float4*>(w_t);
        const float4* b4 = reinterpret_cast<const float4*>(bias);
#pragma unroll
        for (int i = 0; i < LN_NI; ++i) {
            const int c4 = lane + 64 * i;
            h[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c4 < C4) {
                float4 a = b4[c4];
                for (int j = 0; j < k; ++j) {
                    const int tt = t + (j - half) * dil;
                    if (tt >= 0 && tt < Lv) {
                        const float4 xv = x4[(base + tt) * C4 + c4];
                        const float4 wv = w4[(int64_t)j * C4 + c4];
                        a.x = fmaf(wv.x, xv.x, a.x); a.y = fmaf(wv.y, xv.y, a.y);
                        a.z = fmaf(wv.z, xv.z, a.z); a.w = fmaf(wv.w, xv.w, a.w);
                    }
                }
                h[i] = a;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < LN_NI; ++i) {
            const int c4 = lane + 64 * i;
            h[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c4 < C4) h[i] = x4[row * C4 + c4];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_NI; ++i) s += (h[i].x + h[i].y) + (h[i].z + h[i].w);  // slots past C4 hold zeros
    const float mean = wave_sum(s) / (float)C;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < LN_NI; ++i) {
        if (lane + 64 * i < C4) {
            const float dx = h[i].x - mean, dy = h[i].y - mean, dz = h[i].z - mean, dw = h[i].w - mean;
            v += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
    const float rstd = rsqrtf(wave_sum(v) / (float)C + eps);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const float4* bt4 = reinterpret_cast<const float4*>(bt);
#pragma unroll
    for (int i = 0; i < LN_NI; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 < C4) {
            const float4 gg = g4[c4], bb = bt4[c4];
            store4(y + row * C + c4 * 4, live * ((h[i].x - mean) * rstd * gg.x + bb.x), live * ((h[i].y - mean) * rstd * gg.y + bb.y),
                   live * ((h[i].z - mean) * rstd * gg.z + bb.z), live * ((h[i].w - mean) * rstd * gg.w + bb.w));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// dwconv + LayerNorm, fast path: compile-time tap count K, R frames per wavefront, C <= 512.
// Every tap load of every frame is issued before the first FMA (out-of-range taps load a clamped in-range
// frame and are zeroed by a select), so a wave has R*K independent 16-B loads in flight per channel slot
// instead of one dependent load per tap; the tap weights are loaded once and shared by the R frames.
// ---------------------------------------------------------------------------------------------
template <typename OutT, int K, int R>
__global__ __launch_bounds__(256) void dwconv_ln_v2_kernel(const float* __restrict__ x, int64_t M, int L, int C,
                                                           const float* __restrict__ w_t, const float* __restrict__ bias,
                                                           int dil, const float* __restrict__ g,
                                                           const float* __restrict__ bt, float eps, OutT* __restrict__ y,
                                                           const int* __restrict__ seqlen) {
    const int lane = threadIdx.x & 63;
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (r0 >= M) return;  // wave-uniform
    const int C4 = C >> 2;
    constexpr int HALF = (K - 1) / 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float4* w4 = reinterpret_cast<const float4*>(w_t);
    const float4* b4 = reinterpret_cast<const float4*>(bias);
    int tpos[R], lv[R];
    int64_t base[R];
    bool rowok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = r0 + r < M ? r0 + r : M - 1;
        tpos[r] = (int)(row % L);
        base[r] = row - tpos[r];
        lv[r] = seqlen ? seqlen[row / L] : L;
        rowok[r] = r0 + r < M;
    }
    float4 h[R][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c4 = lane + 64 * i;
        const bool act = c4 < C4;
        const int cc = act ? c4 : 0;
        float4 xv[R][K];
        float4 wv[K];
#pragma unroll
        for (int j = 0; j < K; ++j) wv[j] = w4[(int64_t)j * C4 + cc];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int tt = tpos[r] + (j - HALF) * dil;
                const int tc = tt < 0 ? 0 : (tt >= L ? L - 1 : tt);
                xv[r][j] = x4[(base[r] + tc) * C4 + cc];  // clamped in-range load; zeroed below when outside [0, lv)
            }
        const float4 bv = b4[cc];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float4 a = bv;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int tt = tpos[r] + (j - HALF) * dil;
                const float keep = (tt >= 0 && tt < lv[r]) ? 1.f : 0.f;
                a.x = fmaf(wv[j].x * keep, xv[r][j].x, a.x); a.y = fmaf(wv[j].y * keep, xv[r][j].y, a.y);
                a.z = fmaf(wv[j].z * keep, xv[r][j].z, a.z); a.w = fmaf(wv[j].w * keep, xv[r][j].w, a.w);
            }
            h[r][i] = act ? a : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const float4* bt4 = reinterpret_cast<const float4*>(bt);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float s = (h[r][0].x + h[r][0].y) + (h[r][0].z + h[r][0].w) + (h[r][1].x + h[r][1].y) + (h[r][1].z + h[r][1].w);
        const float mean = wave_sum(s) / (float)C;
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (lane + 64 * i < C4) {
                const float dx = h[r][i].x - mean, dy = h[r][i].y - mean, dz = h[r][i].z - mean, dw = h[r][i].w - mean;
                v += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        const float rstd = rsqrtf(wave_sum(v) / (float)C + eps);
        if (!rowok[r]) continue;  // wave-uniform
        const float live = tpos[r] < lv[r] ? 1.f : 0.f;  // rows in the padding are written as zeros
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < C4) {
                const float4 gg = g4[c4], bb = bt4[c4];
                store4(y + (r0 + r) * C + c4 * 4, live * ((h[r][i].x - mean) * rstd * gg.x + bb.x), live * ((h[r][i].y - mean) * rstd * gg.y + bb.y),
                       live * ((h[r][i].z - mean) * rstd * gg.z + bb.z), live * ((h[r][i].w - mean) * rstd * gg.w + bb.w));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// dwconv + LayerNorm v3 ("comb"): a wavefront owns R output frames spaced by the dilation, t_i = t0 + i*dil.  Their
// taps overlap: the R outputs need only R+K-1 distinct input frames (t0 + (q - K/2)*dil), which are loaded once into
// registers as a sliding window — 1 + (K-1)/R loads per output instead of K, for ANY dilation.  Cuts the L2->CU traffic
// of the vocoder's k=7 blocks by 4x at R=8 (HBM already saw each frame once; the re-reads were L2 bandwidth).
// ---------------------------------------------------------------------------------------------
template <typename OutT, int K, int R>
__device__ __forceinline__ void dwconv_ln_v3_body(const float* __restrict__ x, int nseq, int L, int C,
                                                           const float* __restrict__ w_t, const float* __restrict__ bias,
                                                           int dil, int wps /*waves per sequence*/, const float* __restrict__ g,
                                                           const float* __restrict__ bt, float eps, OutT* __restrict__ y,
                                                           const int* __restrict__ seqlen, const int* __restrict__ row_off, int xcd_runs) {
    const int lane = threadIdx.x & 63;
    // Workgroups are dealt to the 8 XCDs round-robin, and each XCD has its own L2: with the plain order the two workgroups that share a halo
    // (neighbours in time) sit on different XCDs and both fetch it over the fabric.  Give each XCD one contiguous run of tiles instead.
    int bid = blockIdx.x;
    if (xcd_runs) {
        const int nb = gridDim.x, q = nb >> 3, rr = nb & 7, xcd = bid & 7, idx = bid >> 3;
        bid = xcd * q + (xcd < rr ? xcd : rr) + idx;
    }
    const int64_t wid = (int64_t)bid * 4 + (threadIdx.x >> 6);
    if (wid >= (int64_t)nseq * wps) return;  // wave-uniform
    const int b = (int)(wid / wps), rem = (int)(wid % wps);
    const int t0 = (rem / dil) * (R * dil) + (rem % dil);
    const int Lv = seqlen ? seqlen[b] : L;  // valid frames of this sequence (<= L)
    constexpr int HALF = (K - 1) / 2, NWIN = R + K - 1;
    const int C4 = C >> 2;
    // packed rows (row_off given): sequence b owns rows row_off[b] .. row_off[b] + seqlen[b]; else b*L .. b*L + L
    const int64_t row0 = row_off ? (int64_t)row_off[b] : (int64_t)b * L;
    if (t0 >= Lv) {  // the whole comb lies in the padding: its rows are defined (zeros) but cost no loads or arithmetic
        if (row_off) return;  // packed layout: there are no padding rows
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int t = t0 + r * dil;
            if (t >= L) break;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (lane + 64 * i < C4) store4(y + (row0 + t) * C + (lane + 64 * i) * 4, 0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const float4* x4 = reinterpret_cast<const float4*>(x) + row0 * C4;
    const float4* w4 = reinterpret_cast<const float4*>(w_t);
    const float4* b4 = reinterpret_cast<const float4*>(bias);
    float4 h[R][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c4 = lane + 64 * i;
        const bool act = c4 < C4;
        const int cc = act ? c4 : 0;
        float4 win[NWIN], wv[K];
#pragma unroll
        for (int j = 0; j < K; ++j) wv[j] = w4[(int64_t)j * C4 + cc];
#pragma unroll
        for (int q = 0; q < NWIN; ++q) {
            const int tt = t0 + (q - HALF) * dil;
            const int hi_ = row_off ? Lv : L;  // clamp inside the rows this sequence owns
            const int tc = tt < 0 ? 0 : (tt >= hi_ ? hi_ - 1 : tt);
            const float4 v = x4[(int64_t)tc * C4 + cc];
            const float keep = (tt >= 0 && tt < Lv) ? 1.f : 0.f;
            win[q] = make_float4(v.x * keep, v.y * keep, v.z * keep, v.w * keep);
        }
        const float4 bv = b4[cc];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float4 a = bv;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                a.x = fmaf(wv[j].x, win[r + j].x, a.x); a.y = fmaf(wv[j].y, win[r + j].y, a.y);
                a.z = fmaf(wv[j].z, win[r + j].z, a.z); a.w = fmaf(wv[j].w, win[r + j].w, a.w);
            }
            h[r][i] = act ? a : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const float4* bt4 = reinterpret_cast<const float4*>(bt);
    float4 gg[2], bb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c4 = lane + 64 * i;
        gg[i] = g4[c4 < C4 ? c4 : 0];
        bb[i] = bt4[c4 < C4 ? c4 : 0];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int t = t0 + r * dil;
        const float s = (h[r][0].x + h[r][0].y) + (h[r][0].z + h[r][0].w) + (h[r][1].x + h[r][1].y) + (h[r][1].z + h[r][1].w);
        const float mean = wave_sum(s) / (float)C;
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (lane + 64 * i < C4) {
                const float dx = h[r][i].x - mean, dy = h[r][i].y - mean, dz = h[r][i].z - mean, dw = h[r][i].w - mean;
                v += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        const float rstd = rsqrtf(wave_sum(v) / (float)C + eps);
        if (t >= L) continue;  // wave-uniform (tail of the sequence)
        if (t >= Lv) {         // padding of a shorter sequence: zeros (rows that do not exist in the packed layout)
            if (row_off) continue;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (lane + 64 * i < C4) store4(y + (row0 + t) * C + (lane + 64 * i) * 4, 0.f, 0.f, 0.f, 0.f);
            continue;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < C4)
                store4(y + (row0 + t) * C + c4 * 4, (h[r][i].x - mean) * rstd * gg[i].x + bb[i].x,
                       (h[r][i].y - mean) * rstd * gg[i].y + bb[i].y, (h[r][i].z - mean) * rstd * gg[i].z + bb[i].z,
                       (h[r][i].w - mean) * rstd * gg[i].w + bb[i].w);
        }
    }
}

template <typename OutT, int K, int R>
__global__ __launch_bounds__(256) void dwconv_ln_v3_kernel(const float* __restrict__ x, int nseq, int L, int C, const float* __restrict__ w_t,
                                                           const float* __restrict__ bias, int dil, int wps, const float* __restrict__ g, const float* __restrict__ bt,
                                                           float eps, OutT* __restrict__ y, const int* __restrict__ seqlen, const int* __restrict__ row_off, int xcd_runs) {
    dwconv_ln_v3_body<OutT, K, R>(x, nseq, L, C, w_t, bias, dil, wps, g, bt, eps, y, seqlen, row_off, xcd_runs);
}
// the same body held to 128 VGPRs (four waves per SIMD instead of three: the vocoder's k = 7 combs of four need 130 by themselves)
template <typename OutT, int K, int R>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void dwconv_ln_v3_occ4_kernel(
    const float* __restrict__ x, int nseq, int L, int C, const float* __restrict__ w_t, const float* __restrict__ bias, int dil, int wps,
    const float* __restrict__ g, const float* __restrict__ bt, float eps, OutT* __restrict__ y, const int* __restrict__ seqlen,
    const int* __restrict__ row_off, int xcd_runs) {
    dwconv_ln_v3_body<OutT, K, R>(x, nseq, L, C, w_t, bias, dil, wps, g, bt, eps, y, seqlen, row_off, xcd_runs);
}

template <typename OutT, int K, int R>
static void launch_dwconv_ln_v3_kr(hipStream_t s, const float* x, int nseq, int L, int C, const float* w_t, const float* bias,
                                   int dil, const float* g, const float* b, float eps, OutT* y, const int* seqlen,
                                   const int* row_off = nullptr) {
    const int wps = ((L + R * dil - 1) / (R * dil)) * dil;
    const int64_t nw = (int64_t)nseq * wps;
    static const int xcd_runs = [] { const char* e = stn::dev_env("STN_DWCONV_XCD"); return e ? atoi(e) : 1; }();  // A/B switch
    // k = 7 combs of four (the vocoder at batch size): 130 VGPRs are three waves per SIMD, 127 are four — 42.5 -> 38.2 us per launch.  (The IEEE-half
    // instantiation needs 184 and would spill: it keeps the plain kernel.)
    if (K == 7 && R == 4 && !std::is_same<OutT, f16_t>::value)
        STN_KLAUNCH((dwconv_ln_v3_occ4_kernel<OutT, K, R>), dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, s, x, nseq, L, C, w_t, bias,
                           dil, wps, g, b, eps, y, seqlen, row_off, xcd_runs);
    else
    STN_KLAUNCH((dwconv_ln_v3_kernel<OutT, K, R>), dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, s, x, nseq, L, C, w_t, bias,
                       dil, wps, g, b, eps, y, seqlen, row_off, xcd_runs);
}

template <typename OutT>
static bool launch_dwconv_ln_v3(hipStream_t s, const float* x, int nseq, int L, int C, const float* w_t, const float* bias,
                                int k, int dil, const float* g, const float* b, float eps, OutT* y, const int* seqlen,
                                const int* row_off = nullptr) {
    if (C > 512 || (k != 5 && k != 7)) return false;
    // enough wavefronts to fill the chip (256 CUs x ~8): long combs only when there are many frames
    const int64_t M = (int64_t)nseq * L;
    if (M >= 32768) {
        // k = 7 (the vocoder, 60 k frames at C3): combs of 4 measured 51 us per launch against 57 us for combs of 8 (twice the wavefronts
        // outweigh 2.5 instead of 1.75 loads per output; combs of 2: 58 us)
        if (k == 5) launch_dwconv_ln_v3_kr<OutT, 5, 8>(s, x, nseq, L, C, w_t, bias, dil, g, b, eps, y, seqlen, row_off);
        else launch_dwconv_ln_v3_kr<OutT, 7, 4>(s, x, nseq, L, C, w_t, bias, dil, g, b, eps, y, seqlen, row_off);
    } else if (M >= 4096) {
        // k = 5 below 16 k frames (the estimator at batch 128: 7.4 k): combs of 2 give twice the wavefronts for 1.5x the loads per output
        // (9.5 -> 8.7 us per launch)
        if (k == 5 && M < 16384) launch_dwconv_ln_v3_kr<OutT, 5, 2>(s, x, nseq, L, C, w_t, bias, dil, g, b, eps, y, seqlen, row_off);
        else if (k == 5) launch_dwconv_ln_v3_kr<OutT, 5, 4>(s, x, nseq, L, C, w_t, bias, dil, g, b, eps, y, seqlen, row_off);
        else launch_dwconv_ln_v3_kr<OutT, 7, 4>(s, x, nseq, L, C, w_t, bias, dil, g, b, eps, y, seqlen, row_off);
    } else if (row_off) {  // the packed layout only exists in this kernel
        // (a single utterance: ~15 wavefronts at combs of 4 — combs of 2 halve the serial work per wavefront)
        if (k == 5 && M < 1024) launch_dwconv_ln_v3_kr<OutT, 5, 2>(s, x, nseq, L, C, w_t, bias, dil, g, b, eps, y, seqlen, row_off);
        else if (k == 5) launch_dwconv_ln_v3_kr<OutT, 5, 4>(s, x, nseq, L, C, w_t, bias, dil, g, b, eps, y, seqlen, row_off);
        else launch_dwconv_ln_v3_kr<OutT, 7, 4>(s, x, nseq, L, C, w_t, bias, dil, g, b, eps, y, seqlen, row_off);
    } else {
        return false;  // few frames: one wave per 2 frames (v2) exposes more parallelism
    }
    return true;
}

template <typename OutT>
static bool launch_dwconv_ln_v2(hipStream_t s, const float* x, int64_t M, int L, int C, const float* w_t, const float* bias,
                                int k, int dil, const float* g, const float* b, float eps, OutT* y, const int* seqlen) {
    constexpr int R = 2;
    if (C > 512 || (k != 5 && k != 7)) return false;
    const dim3 grid((unsigned)((M + 4 * R - 1) / (4 * R)));
    if (k == 5) STN_KLAUNCH((dwconv_ln_v2_kernel<OutT, 5, R>), grid, dim3(256), 0, s, x, M, L, C, w_t, bias, dil, g, b, eps, y, seqlen);
    else STN_KLAUNCH((dwconv_ln_v2_kernel<OutT, 7, R>), grid, dim3(256), 0, s, x, M, L, C, w_t, bias, dil, g, b, eps, y, seqlen);
    return true;
}

static void check_ln_shape(int C) {
    if (C % 4 || C > 4 * 64 * LN_NI) { char m_[256]; snprintf(m_, sizeof m_, "LayerNorm width %d unsupported (C %% 4 == 0, C <= 1024)", C); throw std::invalid_argument(m_); }
}

bool dwconv_ln_supports_packed(int C, int k) { return C <= 512 && C % 4 == 0 && (k == 5 || k == 7); }

void launch_dwconv_ln(hipStream_t s, int out_dtype, const float* x, int B, int L, int C, const float* w_t,
                      const float* bias, int k, int dil, const float* ln_g, const float* ln_b, float eps, void* y,
                      const int* seqlen, const int* row_off) {
    check_ln_shape(C);
    const int64_t M = (int64_t)B * L;
    if (M == 0) return;
    if (row_off && (!seqlen || !dwconv_ln_supports_packed(C, k))) { throw std::invalid_argument("packed dwconv_ln needs lengths, C <= 512, k in {5,7}"); }
    if (out_dtype == BF16 ? launch_dwconv_ln_v3(s, x, B, L, C, w_t, bias, k, dil, ln_g, ln_b, eps, static_cast<uint16_t*>(y), seqlen, row_off)
        : out_dtype == F16 ? launch_dwconv_ln_v3(s, x, B, L, C, w_t, bias, k, dil, ln_g, ln_b, eps, static_cast<f16_t*>(y), seqlen, row_off)
                          : launch_dwconv_ln_v3(s, x, B, L, C, w_t, bias, k, dil, ln_g, ln_b, eps, static_cast<float*>(y), seqlen, row_off))
        return;
    if (out_dtype == BF16 ? launch_dwconv_ln_v2(s, x, M, L, C, w_t, bias, k, dil, ln_g, ln_b, eps, static_cast<uint16_t*>(y), seqlen)
        : out_dtype == F16 ? launch_dwconv_ln_v2(s, x, M, L, C, w_t, bias, k, dil, ln_g, ln_b, eps, static_cast<f16_t*>(y), seqlen)
                          : launch_dwconv_ln_v2(s, x, M, L, C, w_t, bias, k, dil, ln_g, ln_b, eps, static_cast<float*>(y), seqlen))
        return;
    const dim3 grid((unsigned)((M + 3) / 4));
    if (out_dtype == F16)
        STN_KLAUNCH((dwconv_ln_kernel<f16_t, true>), grid, dim3(256), 0, s, x, M, L, C, w_t, bias, k, dil, ln_g,
                           ln_b, eps, static_cast<f16_t*>(y), seqlen);
    else if (out_dtype == BF16)
        STN_KLAUNCH((dwconv_ln_kernel<uint16_t, true>), grid, dim3(256), 0, s, x, M, L, C, w_t, bias, k, dil, ln_g,
                           ln_b, eps, static_cast<uint16_t*>(y), seqlen);
    else
        STN_KLAUNCH((dwconv_ln_kernel<float, true>), grid, dim3(256), 0, s, x, M, L, C, w_t, bias, k, dil, ln_g, ln_b,
                           eps, static_cast<float*>(y), seqlen);
}

void launch_layernorm(hipStream_t s, int out_dtype, const float* x, int64_t M, int C, const float* g, const float* b,
                      float eps, void* y) {
    check_ln_shape(C);
    if (M == 0) return;
    const dim3 grid((unsigned)((M + 3) / 4));
    if (out_dtype == F16)
        STN_KLAUNCH((dwconv_ln_kernel<f16_t, false>), grid, dim3(256), 0, s, x, M, 1, C, nullptr, nullptr, 1, 1, g,
                           b, eps, static_cast<f16_t*>(y), static_cast<const int*>(nullptr));
    else if (out_dtype == BF16)
        STN_KLAUNCH((dwconv_ln_kernel<uint16_t, false>), grid, dim3(256), 0, s, x, M, 1, C, nullptr, nullptr, 1, 1, g,
                           b, eps, static_cast<uint16_t*>(y), static_cast<const int*>(nullptr));
    else
        STN_KLAUNCH((dwconv_ln_kernel<float, false>), grid, dim3(256), 0, s, x, M, 1, C, nullptr, nullptr, 1, 1, g, b,
                           eps, static_cast<float*>(y), static_cast<const int*>(nullptr));
}

// ---------------------------------------------------------------------------------------------
// Folding the pending update of a K4-split launch (kernels_ffn.hip) into the residual stream, inside the kernel that reads x
// next anyway:   x_new = x + gamma * (((p0 + p1) + p2) + p3 + b2) + rowvec[seq]     (fp32, exactly this order in both kernels)
// Everything a thread loads is loaded unconditionally and as whole vectors (a per-element "pointer ? load : constant" makes hipcc
// branch around every load and wait for each one in turn: cdna_hip_programming.md, Projection GEMM item 4(c)); the optional
// time vector is a template parameter, the split count a compile-time constant.
// ---------------------------------------------------------------------------------------------
// fold + LayerNorm, one wavefront per row (rows are independent: x is updated in place)
template <typename OutT, bool F16, bool RV, int S>
__global__ __launch_bounds__(256) void fold_ln_kernel(float* __restrict__ x, int64_t M, int C, const uint16_t* __restrict__ part,
                                                      int64_t pstride, const float* __restrict__ b2, const float* __restrict__ gamma,
                                                      const float* __restrict__ rowvec, int rv_ld, const int* __restrict__ row_b,
                                                      const float* __restrict__ g, const float* __restrict__ bt, float eps, OutT* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;  // wave-uniform
    const int C4 = C >> 2;
    float4* x4 = reinterpret_cast<float4*>(x) + row * C4;
    const float4* rv4 = nullptr;
    if constexpr (RV) rv4 = reinterpret_cast<const float4*>(rowvec + (size_t)(row_b ? row_b[row] : 0) * rv_ld);
    const float4* b24 = reinterpret_cast<const float4*>(b2);
    const float4* gm4 = reinterpret_cast<const float4*>(gamma);
    float4 h[LN_NI];
#pragma unroll
    for (int i = 0; i < LN_NI; ++i) {
        const int c4 = lane + 64 * i;
        h[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c4 < C4) {
            const float4 xo = x4[c4];
            const float4 bb = b24[c4], gm = gm4[c4];
            float4 tv = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (RV) tv = rv4[c4];
            float acc[4];
            fold_sum<F16, 4, S>(part + (size_t)row * C + c4 * 4, pstride, acc);
            h[i] = fold_four(xo, acc, bb, gm, tv);
            x4[c4] = h[i];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_NI; ++i) s += (h[i].x + h[i].y) + (h[i].z + h[i].w);
    const float mean = wave_sum(s) / (float)C;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < LN_NI; ++i)
        if (lane + 64 * i < C4) {
            const float dx = h[i].x - mean, dy = h[i].y - mean, dz = h[i].z - mean, dw = h[i].w - mean;
            v += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    const float rstd = rsqrtf(wave_sum(v) / (float)C + eps);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const float4* bt4 = reinterpret_cast<const float4*>(bt);
#pragma unroll
    for (int i = 0; i < LN_NI; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 < C4) {
            const float4 gg = g4[c4], bb = bt4[c4];
            store4(y + row * C + c4 * 4, (h[i].x - mean) * rstd * gg.x + bb.x, (h[i].y - mean) * rstd * gg.y + bb.y,
                   (h[i].z - mean) * rstd * gg.z + bb.z, (h[i].w - mean) * rstd * gg.w + bb.w);
        }
    }
}

static void check_fold_args(const FoldArgs& f, int64_t M, int C, const char* who) {
    if (!f.part || (f.S != 4 && f.S != 8 && f.S != 12 && f.S != 24) || f.part_stride < M * C || !f.b2 || !f.gamma || (f.rowvec && f.rv_ld % 4) || (reinterpret_cast<uintptr_t>(f.part) & 15) ||
        (reinterpret_cast<uintptr_t>(f.b2) & 15) || (reinterpret_cast<uintptr_t>(f.gamma) & 15) || (f.rowvec && (reinterpret_cast<uintptr_t>(f.rowvec) & 15)))
        throw std::invalid_argument(std::string(who) + ": needs 16-byte aligned partial sums of 4, 8, 12 or 24 splits, b2 and gamma");
}

void launch_fold_ln(hipStream_t s, int act_dtype, float* x, int64_t M, int C, const FoldArgs& f, const float* g, const float* b, float eps, void* y) {
    check_ln_shape(C);
    if (M == 0) return;
    if (!is_half(act_dtype)) throw std::invalid_argument("launch_fold_ln: 16-bit activation format needed");
    check_fold_args(f, M, C, "launch_fold_ln");
    const dim3 grid((unsigned)((M + 3) / 4));
    const uint16_t* P = static_cast<const uint16_t*>(f.part);
#define STN_FOLD_LN(OUT, F16_, RV_, S_) STN_KLAUNCH((fold_ln_kernel<OUT, F16_, RV_, S_>), grid, dim3(256), 0, s, x, M, C, P, f.part_stride, f.b2, f.gamma, f.rowvec, \
                                                    f.rv_ld, f.row_b, g, b, eps, static_cast<OUT*>(y))
#define STN_FOLD_LN_S(OUT, F16_, RV_) do { if (f.S == 4) STN_FOLD_LN(OUT, F16_, RV_, 4); else if (f.S == 8) STN_FOLD_LN(OUT, F16_, RV_, 8); else if (f.S == 12) STN_FOLD_LN(OUT, F16_, RV_, 12); else STN_FOLD_LN(OUT, F16_, RV_, 24); } while (0)
    if (act_dtype == F16) { if (f.rowvec) STN_FOLD_LN_S(f16_t, true, true); else STN_FOLD_LN_S(f16_t, true, false); }
    else { if (f.rowvec) STN_FOLD_LN_S(uint16_t, false, true); else STN_FOLD_LN_S(uint16_t, false, false); }
#undef STN_FOLD_LN_S
#undef STN_FOLD_LN
}

// fold + depthwise conv + LayerNorm on packed rows.  A workgroup (16 wavefronts) owns a run of consecutive frames of ONE sequence
// (a sequence is cut into ceil(len / 32) runs of equal length): it folds those frames and the (K-1)/2 * dil halo frames on either
// side into an fp32 LDS image (phase 1), runs the conv + LayerNorm out of LDS, one frame per half wavefront (phase 2), and writes
// the folded frames it owns to x_out.  x_out != x_in: the halo frames are folded again by the neighbouring workgroup from the
// same inputs.
// The kernel is a chain of memory latencies, not of bytes (one workgroup per CU, ~150 KB each): everything is arranged so that
// the chain is ONE global round trip long — the conv / LayerNorm parameters ride to LDS beside the phase-1 loads (no global load
// behind the barrier), nothing is stored to global memory before the barrier (a store in flight would be waited for there), and
// the x_out stores are the last thing a thread issues.
// Few sequences (a single utterance: two workgroups) are cut into runs of 8 frames instead: more workgroups, one pass of phase-1 loads each
// instead of two to four dependent ones; a frame's arithmetic does not depend on the run it falls in, so the result is the same bit for bit.
static constexpr int FOLD_TCH = 32, FOLD_TCH_FEW = 8, FOLD_TCH_MAX = 48, FOLD_NT = 1024;  // FOLD_TCH == 2 * wavefronts per workgroup (the longest run)
template <typename OutT, bool F16, int K, bool RV, int FOLD_NSLOT /* float4 slots per lane of a half wavefront: ceil(C / 128) */, int S>
__global__ __launch_bounds__(FOLD_NT) void fold_dwconv_ln_kernel(const float* __restrict__ xin, float* __restrict__ xout, int cps, int C,
                                                                 const uint16_t* __restrict__ part, int64_t pstride,
                                                                 const float* __restrict__ b2, const float* __restrict__ gamma,
                                                                 const float* __restrict__ rowvec, int rv_ld, const float* __restrict__ w_t,
                                                                 const float* __restrict__ bias, int dil, const float* __restrict__ g,
                                                                 const float* __restrict__ bt, float eps, float inv_c, OutT* __restrict__ y,
                                                                 const int* __restrict__ seqlen, const int* __restrict__ row_off,
                                                                 unsigned long long* __restrict__ ts, int tch /* frames per run: FOLD_TCH or FOLD_TCH_FEW */) {
    extern __shared__ __attribute__((aligned(16))) float fold_sm[];
    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if (ts) st0 = __builtin_readcyclecounter();
    const int b = (int)blockIdx.x / cps, c = (int)blockIdx.x % cps;
    const int Lv = seqlen[b];
    const int64_t row0 = (int64_t)row_off[b];
    const int nch = (Lv + tch - 1) / tch;
    if (c >= nch) return;
    const int per = (Lv + nch - 1) / nch;
    const int t0 = c * per, t1 = min(t0 + per, Lv);
    if (t0 >= t1) return;
    constexpr int HALF = (K - 1) / 2;
    const int w0 = max(t0 - HALF * dil, 0), w1 = min(t1 + HALF * dil, Lv), nw = w1 - w0;
    const int tid = threadIdx.x, C8 = C >> 3, C4 = C >> 2;
    // the parameters of phase 2 as one LDS block behind the image: [K taps][C] | conv bias | LayerNorm g | LayerNorm b
    float* const zrow = fold_sm + (size_t)(tch + (K - 1) * dil) * C;  // a row of zeros: what a tap outside the sequence reads
    float* const wsm = zrow + C;
    constexpr int NPV = (K + 3 + 7) / 8;  // float4 per thread: (K + 3) * C4 <= NPV * FOLD_NT for C <= 512, K <= 7 ... checked by the launcher
    float4 pv[NPV];
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
        const int q = tid + i * FOLD_NT;  // float4 index into the block
        const int seg = q / C4, c4 = q - seg * C4;
        const float* src = seg < K ? w_t + (size_t)seg * C : seg == K ? bias : seg == K + 1 ? g : bt;
        pv[i] = q < (K + 3) * C4 ? reinterpret_cast<const float4*>(src)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // ---- phase 1: a thread keeps ONE 8-channel group (its b2 / gamma / time-vector values are loaded once) and walks down the
    // window rows, `rpp` rows apart; every load of U rows is issued before the first use ----
    // Three rows per thread and trip when there are four partial sums: a run of 29 frames with a halo of 2 x 16 (dilation 8) is 61 rows = ONE trip of
    // 3 x 21 rows — one global round trip instead of two dependent ones.  The third row's loads are issued only by the threads that have one.
    // (the wider variants — C = 512, seven taps — keep two rows: three would spill; more splits: one row at a time, the partial sums in chunks of 12 loads)
    constexpr int U = S <= 4 ? (K == 5 && FOLD_NSLOT <= 3 ? 3 : 2) : 1;
    const int rpp = FOLD_NT / C8;  // rows per pass of the workgroup
    const int c8 = tid % C8, rq = tid / C8;
    if (rq < rpp) {
        const float4* bp = reinterpret_cast<const float4*>(b2 + c8 * 8);
        const float4* gp = reinterpret_cast<const float4*>(gamma + c8 * 8);
        const float4 bb0 = bp[0], bb1 = bp[1], gm0 = gp[0], gm1 = gp[1];
        float4 tv0 = make_float4(0.f, 0.f, 0.f, 0.f), tv1 = tv0;
        if constexpr (RV) { const float4* tp = reinterpret_cast<const float4*>(rowvec + (size_t)b * rv_ld + c8 * 8); tv0 = tp[0]; tv1 = tp[1]; }
        for (int r0 = rq; r0 < nw; r0 += rpp * U) {
            float4 xa[U][2];
            float acc[U][8];
            int64_t mrow[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = min(r0 + u * rpp, nw - 1);  // (past the end: the last row again, stored nowhere)
                mrow[u] = row0 + w0 + r;
            }
            if constexpr (U >= 2) {  // every load of all rows is issued before the first use
                constexpr int CH = S;
                uint4 w[U][CH];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float4* xp = reinterpret_cast<const float4*>(xin + mrow[u] * C + c8 * 8);
                    xa[u][0] = xp[0]; xa[u][1] = xp[1];
#pragma unroll
                    for (int sp = 0; sp < CH; ++sp) w[u][sp] = *reinterpret_cast<const uint4*>(part + (size_t)sp * pstride + (size_t)mrow[u] * C + c8 * 8);
                }
                if constexpr (U == 3) {
                    if (r0 + 2 * rpp < nw) {
                        const float4* xp = reinterpret_cast<const float4*>(xin + mrow[2] * C + c8 * 8);
                        xa[2][0] = xp[0]; xa[2][1] = xp[1];
#pragma unroll
                        for (int sp = 0; sp < CH; ++sp) w[2][sp] = *reinterpret_cast<const uint4*>(part + (size_t)sp * pstride + (size_t)mrow[2] * C + c8 * 8);
                    } else {
                        xa[2][0] = xa[2][1] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int sp = 0; sp < CH; ++sp) w[2][sp] = make_uint4(0u, 0u, 0u, 0u);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int sp = 0; sp < CH; ++sp) {
                        const unsigned wj[4] = {w[u][sp].x, w[u][sp].y, w[u][sp].z, w[u][sp].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float lo = p16_to_f<F16>(wj[j] & 0xFFFFu), hi = p16_to_f<F16>(wj[j] >> 16);
                            if (sp == 0) { acc[u][2 * j] = lo; acc[u][2 * j + 1] = hi; }
                            else { acc[u][2 * j] += lo; acc[u][2 * j + 1] += hi; }
                        }
                    }
            } else {
                const float4* xp = reinterpret_cast<const float4*>(xin + mrow[0] * C + c8 * 8);
                xa[0][0] = xp[0]; xa[0][1] = xp[1];
                fold_sum<F16, 8, S>(part + (size_t)mrow[0] * C + c8 * 8, pstride, acc[0]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + u * rpp;
                if (r < nw) {
                    float4* sp4 = reinterpret_cast<float4*>(fold_sm + (size_t)r * C + c8 * 8);
                    sp4[0] = fold_four(xa[u][0], acc[u], bb0, gm0, tv0);
                    sp4[1] = fold_four(xa[u][1], acc[u] + 4, bb1, gm1, tv1);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
        const int q = tid + i * FOLD_NT;
        if (q < (K + 3) * C4) reinterpret_cast<float4*>(wsm)[q] = pv[i];
    }
    if (tid < C4) reinterpret_cast<float4*>(zrow)[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ts) st1 = __builtin_readcyclecounter();
    __syncthreads();
    if (ts) st2 = __builtin_readcyclecounter();
    // ---- phase 2: one frame per HALF wavefront (32 lanes x FOLD_NSLOT float4 slots cover C): all of a run's <= 32 frames are done
    // in one pass of the 16 wavefronts, and the two LayerNorm reductions are 4 DPP steps + one swizzle over 32 lanes ----
    const int lane = tid & 63, l32 = lane & 31;
    const float4* ws4 = reinterpret_cast<const float4*>(wsm);
    for (int tp = t0; tp < t1; tp += FOLD_TCH) {  // (a run longer than 32 frames — run_frames 40 / 48 — takes a second pass)
    asm volatile("" ::: "memory");  // (keeps the pass's LDS reads of the taps and LayerNorm parameters inside it: hoisted, they would spill)
    const int t = tp + 2 * (tid >> 6) + (lane >> 5);
    const bool live = t < t1;
    const int tl = live ? t : t0;  // (a half wavefront beyond the run computes frame t0 again and stores nothing)
    float4 h[FOLD_NSLOT];
#pragma unroll
    for (int i = 0; i < FOLD_NSLOT; ++i) {
        const int c4 = l32 + 32 * i, cq = c4 < C4 ? c4 : 0;
        float4 a = ws4[K * C4 + cq];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int tt = tl + (j - HALF) * dil;
            const bool in = tt >= 0 && tt < Lv;  // (then w0 <= tt < w1)
            const float4 xv = *reinterpret_cast<const float4*>((in ? fold_sm + (size_t)(tt - w0) * C : zrow) + cq * 4);
            const float4 wv = ws4[j * C4 + cq];
            a.x = fmaf(wv.x, xv.x, a.x); a.y = fmaf(wv.y, xv.y, a.y);
            a.z = fmaf(wv.z, xv.z, a.z); a.w = fmaf(wv.w, xv.w, a.w);
        }
        h[i] = c4 < C4 ? a : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < FOLD_NSLOT; ++i) s += (h[i].x + h[i].y) + (h[i].z + h[i].w);
    const float mean = half_wave_sum(s) * inv_c;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < FOLD_NSLOT; ++i)
        if (l32 + 32 * i < C4) {
            const float dx = h[i].x - mean, dy = h[i].y - mean, dz = h[i].z - mean, dw = h[i].w - mean;
            v += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    const float rstd = rsqrtf(half_wave_sum(v) * inv_c + eps);
    if (live) {
#pragma unroll
        for (int i = 0; i < FOLD_NSLOT; ++i) {
            const int c4 = l32 + 32 * i;
            if (c4 < C4) {
                const float4 gg = ws4[(K + 1) * C4 + c4], bb = ws4[(K + 2) * C4 + c4];
                store4(y + (row0 + t) * C + c4 * 4, (h[i].x - mean) * rstd * gg.x + bb.x, (h[i].y - mean) * rstd * gg.y + bb.y,
                       (h[i].z - mean) * rstd * gg.z + bb.z, (h[i].w - mean) * rstd * gg.w + bb.w);
            }
        }
    }
    }
    // ---- the folded frames this run owns -> x_out (each thread: the image rows it wrote itself) ----
    if (rq < rpp)
        for (int r = rq; r < nw; r += rpp) {
            const int tt = w0 + r;
            if (tt >= t0 && tt < t1) {
                const float4* sp4 = reinterpret_cast<const float4*>(fold_sm + (size_t)r * C + c8 * 8);
                float4* op = reinterpret_cast<float4*>(xout + (row0 + tt) * C + c8 * 8);
                op[0] = sp4[0]; op[1] = sp4[1];
            }
        }
    if (ts && tid == 0) {
        unsigned long long* tp = ts + (size_t)blockIdx.x * 4;
        tp[0] = st0; tp[1] = st1; tp[2] = st2; tp[3] = __builtin_readcyclecounter();
    }
}

int fold_run_frames(const int* lengths, int B, int n_cu) {
    if (!lengths || B <= 0 || n_cu <= 0) return 0;
    // The grid is B x ceil(Lmax / run): the runs a sequence does not have are workgroups too — they leave at once, but each takes a CU's LDS and 16
    // wave slots on the way (measured: 238 real workgroups in a grid of 336 run at the two-round time, 16.2 us; the same batch as 224 of 224: 13.2).
    // So the grid, not the number of real runs, is what is kept within whole rounds.
    int Lmax = 0;
    for (int i = 0; i < B; ++i) Lmax = std::max(Lmax, lengths[i]);
    // Only the clear case is acted on: a longer run that brings the whole grid into ONE round.  With several rounds either way the count of rounds stops
    // predicting the time (mixed lengths, 128 sequences of up to ~250 frames: runs of 48 = 3 rounds of grid measured 40.2 us against 37.6 for runs of
    // 32 = 4 rounds, half of them placeholders).
    if ((long)B * ((Lmax + FOLD_TCH - 1) / FOLD_TCH) <= n_cu) return 0;
    for (int tch = FOLD_TCH + 8; tch <= FOLD_TCH_MAX; tch += 8)
        if ((long)B * ((Lmax + tch - 1) / tch) <= n_cu) return tch;
    return 0;
}

static size_t fold_dwconv_lds(int C, int k, int dil, int tch = FOLD_TCH) { return ((size_t)(tch + (k - 1) * dil) + 1 + (size_t)(k + 3)) * C * 4; }  // image + zero row + parameter block
bool fold_dwconv_ln_supported(int C, int k, int dil) {
    return C % 8 == 0 && C <= 512 && (k == 5 || k == 7) && dil >= 1 && fold_dwconv_lds(C, k, dil) <= 160 * 1024;
}

template <typename OutT, bool F16, int K, bool RV, int NSLOT, int S>
static void launch_fold_dwconv_ln_t3(hipStream_t s, const float* x_in, float* x_out, int B, int L, int C, const FoldArgs& f, const float* w_t,
                                     const float* bias, int dil, const float* g, const float* b, float eps, OutT* y, const int* seqlen, const int* row_off) {
    static PerDeviceOnce attr_once;
    if (attr_once.need())
        stn_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&fold_dwconv_ln_kernel<OutT, F16, K, RV, NSLOT, S>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          160 * 1024), "hipFuncSetAttribute(fold_dwconv_ln)");
    static const int force = [] { const char* e = stn::dev_env("STN_FOLD_TCH"); return e ? atoi(e) : 0; }();  // A/B switch: 8, 32, 40 or 48
    int tch = force == FOLD_TCH || force == FOLD_TCH_FEW ? force : (int64_t)B * ((L + FOLD_TCH - 1) / FOLD_TCH) < 64 ? FOLD_TCH_FEW : FOLD_TCH;
    if ((force == 40 || force == 48) && fold_dwconv_lds(C, K, dil, force) <= 160 * 1024) tch = force;
    if (!force && tch == FOLD_TCH && f.run_frames > FOLD_TCH && f.run_frames <= FOLD_TCH_MAX && f.run_frames % 8 == 0 && fold_dwconv_lds(C, K, dil, f.run_frames) <= 160 * 1024)
        tch = f.run_frames;  // fewer rounds of workgroups for these lengths (fold_run_frames)
    const int cps = (L + tch - 1) / tch;
    STN_KLAUNCH((fold_dwconv_ln_kernel<OutT, F16, K, RV, NSLOT, S>), dim3((unsigned)((int64_t)B * cps)), dim3(FOLD_NT), fold_dwconv_lds(C, K, dil, tch), s, x_in, x_out, cps, C,
                static_cast<const uint16_t*>(f.part), f.part_stride, f.b2, f.gamma, f.rowvec, f.rv_ld, w_t, bias, dil, g, b, eps, 1.0f / (float)C, y, seqlen, row_off, f.ts, tch);
}
template <typename OutT, bool F16, int K, bool RV, int NSLOT>
static void launch_fold_dwconv_ln_t2(hipStream_t s, const float* x_in, float* x_out, int B, int L, int C, const FoldArgs& f, const float* w_t,
                                     const float* bias, int dil, const float* g, const float* b, float eps, OutT* y, const int* seqlen, const int* row_off) {
    if (f.S == 4) launch_fold_dwconv_ln_t3<OutT, F16, K, RV, NSLOT, 4>(s, x_in, x_out, B, L, C, f, w_t, bias, dil, g, b, eps, y, seqlen, row_off);
    else if (f.S == 8) launch_fold_dwconv_ln_t3<OutT, F16, K, RV, NSLOT, 8>(s, x_in, x_out, B, L, C, f, w_t, bias, dil, g, b, eps, y, seqlen, row_off);
    else if (f.S == 12) launch_fold_dwconv_ln_t3<OutT, F16, K, RV, NSLOT, 12>(s, x_in, x_out, B, L, C, f, w_t, bias, dil, g, b, eps, y, seqlen, row_off);
    else launch_fold_dwconv_ln_t3<OutT, F16, K, RV, NSLOT, 24>(s, x_in, x_out, B, L, C, f, w_t, bias, dil, g, b, eps, y, seqlen, row_off);
}
template <typename OutT, bool F16, int K, bool RV>
static void launch_fold_dwconv_ln_t(hipStream_t s, const float* x_in, float* x_out, int B, int L, int C, const FoldArgs& f, const float* w_t,
                                    const float* bias, int dil, const float* g, const float* b, float eps, OutT* y, const int* seqlen, const int* row_off) {
    if (C <= 384) launch_fold_dwconv_ln_t2<OutT, F16, K, RV, 3>(s, x_in, x_out, B, L, C, f, w_t, bias, dil, g, b, eps, y, seqlen, row_off);
    else launch_fold_dwconv_ln_t2<OutT, F16, K, RV, 4>(s, x_in, x_out, B, L, C, f, w_t, bias, dil, g, b, eps, y, seqlen, row_off);
}

void launch_fold_dwconv_ln(hipStream_t s, int act_dtype, const float* x_in, float* x_out, int B, int L, int C, const FoldArgs& f, const float* w_t,
                           const float* bias, int k, int dil, const float* ln_g, const float* ln_b, float eps, void* y, const int* seqlen,
                           const int* row_off) {
    if (B == 0 || L == 0) return;
    if (!is_half(act_dtype) || !seqlen || !row_off || x_in == x_out || !fold_dwconv_ln_supported(C, k, dil) ||
        (int64_t)B * ((L + FOLD_TCH_FEW - 1) / FOLD_TCH_FEW) > 0x7FFFFFFFll)
        throw std::invalid_argument("launch_fold_dwconv_ln: packed 16-bit rows, separate output, k in {5,7}, C % 8 == 0, C <= 512 needed");
    check_fold_args(f, 0, C, "launch_fold_dwconv_ln");
#define STN_FOLD_DW(OUT, F16_, K_) do { if (f.rowvec) launch_fold_dwconv_ln_t<OUT, F16_, K_, true>(s, x_in, x_out, B, L, C, f, w_t, bias, dil, ln_g, ln_b, eps, static_cast<OUT*>(y), seqlen, row_off); \
                                        else launch_fold_dwconv_ln_t<OUT, F16_, K_, false>(s, x_in, x_out, B, L, C, f, w_t, bias, dil, ln_g, ln_b, eps, static_cast<OUT*>(y), seqlen, row_off); } while (0)
    if (act_dtype == F16) { if (k == 5) STN_FOLD_DW(f16_t, true, 5); else STN_FOLD_DW(f16_t, true, 7); }
    else { if (k == 5) STN_FOLD_DW(uint16_t, false, 5); else STN_FOLD_DW(uint16_t, false, 7); }
#undef STN_FOLD_DW
}

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
__global__ void embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ emb, int vocab, int L, int C,
                             const int* __restrict__ len, float* __restrict__ x, const int* __restrict__ row_off) {
    const int row = blockIdx.x, b = row / L, t = row - b * L;
    if (row_off && t >= len[b]) return;  // packed destination: the position does not exist
    const int64_t id = ids[row];
    const bool ok = t < len[b] && id >= 0 && id < vocab;
    const int C4 = C >> 2;
    float4* o = reinterpret_cast<float4*>(x) + (row_off ? (int64_t)row_off[b] + t : (int64_t)row) * C4;
    const float4* e = reinterpret_cast<const float4*>(emb) + (ok ? id : 0) * C4;
    for (int c = threadIdx.x; c < C4; c += blockDim.x) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = e[c];
        o[c] = v;
    }
}
void launch_embed(hipStream_t s, const int64_t* ids, const float* emb, int vocab, int B, int L, int C, const int* len,
                  float* x, const int* row_off) {
    if (B * L == 0) return;
    STN_KLAUNCH(embed_kernel, dim3(B * L), dim3(64), 0, s, ids, emb, vocab, L, C, len, x, row_off);
}

__global__ void mask_to_len_kernel(const float* __restrict__ mask, int L, int* __restrict__ len) {
    const int b = blockIdx.x;
    float c = 0.f;
    for (int t = threadIdx.x; t < L; t += 64) c += mask[(int64_t)b * L + t] > 0.5f ? 1.f : 0.f;
    c = wave_sum(c);
    if (threadIdx.x == 0) len[b] = (int)(c + 0.5f);
}
void launch_mask_to_len(hipStream_t s, const float* mask, int B, int L, int* len) {
    if (B == 0) return;
    STN_KLAUNCH(mask_to_len_kernel, dim3(B), dim3(64), 0, s, mask, L, len);
}

template <typename OutT>
__global__ void ncl_to_rows_kernel(const float* __restrict__ in, int C, int L, int ldo, int64_t n, OutT* __restrict__ out,
                                   const int* __restrict__ len, const int* __restrict__ row_off) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over [B][L][ldo]
    if (i >= n) return;
    const int c = (int)(i % ldo);
    const int64_t r = i / ldo;
    const int t = (int)(r % L);
    const int64_t b = r / L;
    const float v = c < C ? in[(b * C + c) * L + t] : 0.f;
    if (row_off) {  // packed destination: only the frames the sequence owns exist
        if (t < len[b]) store1(out + ((int64_t)row_off[b] + t) * ldo + c, v);
    } else {
        store1(out + i, v);
    }
}
void launch_ncl_to_rows(hipStream_t s, int out_dtype, const float* in, int B, int C, int L, void* out, int ld_out, const int* len,
                        const int* row_off) {
    const int ldo = ld_out > 0 ? ld_out : C;
    const int64_t n = (int64_t)B * ldo * L;
    if (n == 0) return;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (out_dtype == F16) STN_KLAUNCH(ncl_to_rows_kernel<f16_t>, grid, dim3(256), 0, s, in, C, L, ldo, n, static_cast<f16_t*>(out), len, row_off);
    else if (out_dtype == BF16) STN_KLAUNCH(ncl_to_rows_kernel<uint16_t>, grid, dim3(256), 0, s, in, C, L, ldo, n, static_cast<uint16_t*>(out), len, row_off);
    else STN_KLAUNCH(ncl_to_rows_kernel<float>, grid, dim3(256), 0, s, in, C, L, ldo, n, static_cast<float*>(out), len, row_off);
}

// 32 x 32 LDS tile transpose: reads of v are coalesced along d, writes of out along t.  ZT != void: the new latent is ALSO written as the rows the
// next step's input projection reads (z[row][d], row stride ldz, the activation format — what launch_ncl_to_rows would make of `out`, bit for bit), through
// a second transpose of the tile: the step after this one starts without that launch (its strided reads cost 11 us for 6 MB).
template <typename ZT>
__global__ __launch_bounds__(256) void euler_ncl_kernel(const float* __restrict__ prev, const float* __restrict__ v,
                                                        const float* __restrict__ dt, const int* __restrict__ len, int D, int L,
                                                        float* __restrict__ out, const int* __restrict__ row_off, ZT* __restrict__ z, int ldz) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int n = len ? len[b] : L;
    const int64_t vrow0 = row_off ? (int64_t)row_off[b] : (int64_t)b * L;
    const int vrows = row_off ? n : L;  // rows of v this sequence owns
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = t0 + ty + 8 * k, d = d0 + tx;
        tile[ty + 8 * k][tx] = (t < vrows && d < D) ? v[(vrow0 + t) * D + d] : 0.f;
    }
    __syncthreads();
    const float scale = dt[b];
    float nv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int d = d0 + ty + 8 * k, t = t0 + tx;
        nv[k] = 0.f;
        if (d < D && t < L) {
            const int64_t o = ((int64_t)b * D + d) * L + t;
            nv[k] = t < n ? prev[o] + tile[tx][ty + 8 * k] * scale : 0.f;
            out[o] = nv[k];
        }
    }
    if constexpr (!std::is_same<ZT, void>::value) {
        __syncthreads();  // every read of the velocity tile is done
#pragma unroll
        for (int k = 0; k < 4; ++k) tile[ty + 8 * k][tx] = nv[k];  // [d][t]
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = t0 + ty + 8 * k, d = d0 + tx;
            if (t < vrows && d < D) store1(z + (vrow0 + t) * ldz + d, tile[tx][ty + 8 * k]);
        }
    }
}
void launch_euler_ncl(hipStream_t s, const float* prev, const float* v, const float* dt, const int* len, int B, int D, int L, float* out,
                      const int* row_off, void* z_rows, int z_dtype, int ldz) {
    if (B * D * L == 0) return;
    if (row_off && !len) { throw std::invalid_argument("packed euler_ncl needs lengths"); }
    const dim3 grid((L + 31) / 32, (D + 31) / 32, B);
    if (!z_rows) STN_KLAUNCH(euler_ncl_kernel<void>, grid, dim3(256), 0, s, prev, v, dt, len, D, L, out, row_off, static_cast<void*>(nullptr), 0);
    else if (z_dtype == F16) STN_KLAUNCH(euler_ncl_kernel<f16_t>, grid, dim3(256), 0, s, prev, v, dt, len, D, L, out, row_off, static_cast<f16_t*>(z_rows), ldz);
    else if (z_dtype == BF16) STN_KLAUNCH(euler_ncl_kernel<uint16_t>, grid, dim3(256), 0, s, prev, v, dt, len, D, L, out, row_off, static_cast<uint16_t*>(z_rows), ldz);
    else STN_KLAUNCH(euler_ncl_kernel<float>, grid, dim3(256), 0, s, prev, v, dt, len, D, L, out, row_off, static_cast<float*>(z_rows), ldz);
}

// row_off[b] = sum of len[0..b) (row_off[B] = total), row_b[row_off[b] + t] = b: the packed-row bookkeeping, one block
__global__ void row_map_kernel(const int* __restrict__ len, int B, int* __restrict__ row_off, int* __restrict__ row_b, int rows_padded) {
    __shared__ int off_s[1025];
    // exclusive prefix sum of up to 1024 lengths: every thread loads one, Hillis-Steele over LDS (10 steps), instead of one thread walking them
    const int tid = threadIdx.x;
    int v = tid < B ? len[tid] : 0;
    off_s[tid + 1] = v;
    if (tid == 0) off_s[0] = 0;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int add = tid + 1 > d ? off_s[tid + 1 - d] : 0;
        __syncthreads();
        if (tid + 1 > d) off_s[tid + 1] += add;
        __syncthreads();
    }
    for (int b = tid; b <= B; b += blockDim.x) row_off[b] = off_s[b];
    if (!row_b) return;
    // row -> sequence: every thread walks its own rows and finds the sequence by bisection over the offsets (B <= 1024: 10 steps)
    const int total = off_s[B];
    for (int r = tid; r < total; r += blockDim.x) {
        int lo = 0, hi = B;  // off_s[lo] <= r < off_s[hi]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off_s[mid] <= r) lo = mid; else hi = mid; }
        row_b[r] = lo;
    }
    // dead rows behind the last sequence (a row count rounded up to a shape bucket): sequence 0, so that per-row lookups stay in range
    for (int t = total + tid; t < rows_padded; t += blockDim.x) row_b[t] = 0;
}

// packed rows [sum len][W] -> padded [B][T][W] with zeros past each sequence's length (W % 4 == 0)
__global__ void unpack_rows_kernel(const float* __restrict__ src, const int* __restrict__ len, const int* __restrict__ row_off, int T,
                                   int W4, int64_t n4, float* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over [B][T][W/4]
    if (i >= n4) return;
    const int c = (int)(i % W4);
    const int64_t r = i / W4;
    const int t = (int)(r % T);
    const int b = (int)(r / T);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < len[b]) v = reinterpret_cast<const float4*>(src)[((int64_t)row_off[b] + t) * W4 + c];
    reinterpret_cast<float4*>(dst)[i] = v;
}
void launch_unpack_rows(hipStream_t s, const float* src, const int* len, const int* row_off, int B, int T, int W, float* dst) {
    const int64_t n4 = (int64_t)B * T * (W / 4);
    if (n4 == 0) return;
    if (W % 4) { throw std::invalid_argument("unpack_rows needs W % 4 == 0"); }
    STN_KLAUNCH(unpack_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, src, len, row_off, T, W / 4, n4, dst);
}
void launch_row_map(hipStream_t s, const int* len, int B, int* row_off, int* row_b, int rows_padded) {
    if (B <= 0) return;
    if (B > 1024) { throw std::invalid_argument("packed layout supports at most 1024 sequences per batch"); }
    STN_KLAUNCH(row_map_kernel, dim3(1), dim3(1024), 0, s, len, B, row_off, row_b, rows_padded);
}

template <typename OutT>
__global__ void cast_kernel(const float* __restrict__ in, int64_t n, OutT* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) store1(out + i, in[i]);
}
void launch_cast(hipStream_t s, int out_dtype, const float* in, int64_t n, void* out) {
    if (n == 0) return;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (out_dtype == F16) STN_KLAUNCH(cast_kernel<f16_t>, grid, dim3(256), 0, s, in, n, static_cast<f16_t*>(out));
    else if (out_dtype == BF16) STN_KLAUNCH(cast_kernel<uint16_t>, grid, dim3(256), 0, s, in, n, static_cast<uint16_t*>(out));
    else STN_KLAUNCH(cast_kernel<float>, grid, dim3(256), 0, s, in, n, static_cast<float*>(out));
}

__global__ void add_rowvec_kernel(float* __restrict__ x, const float* __restrict__ v, int ldv, int L, int C4, int64_t n4,
                                  const int* __restrict__ len) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over [B*L][C/4]
    if (i >= n4) return;
    const int c4 = (int)(i % C4);
    const int64_t r = i / C4;
    const int t = (int)(r % L);
    const int b = (int)(r / L);
    if (len && t >= len[b]) return;
    float4* xp = reinterpret_cast<float4*>(x) + i;
    const float4 a = *xp, d = *reinterpret_cast<const float4*>(v + (int64_t)b * ldv + c4 * 4);
    *xp = make_float4(a.x + d.x, a.y + d.y, a.z + d.z, a.w + d.w);
}
void launch_add_rowvec(hipStream_t s, float* x, const float* v, int ldv, int B, int L, int C, const int* len) {
    const int64_t n4 = (int64_t)B * L * (C / 4);
    if (n4 == 0) return;
    if (C % 4 || ldv % 4) { throw std::invalid_argument("add_rowvec needs C % 4 == 0"); }
    STN_KLAUNCH(add_rowvec_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, x, v, ldv, L, C / 4, n4, len);
}

__global__ void time_embed_kernel(const float* __restrict__ cur, const float* __restrict__ tot, int dim, float scale,
                                  float* __restrict__ te) {
    const int b = blockIdx.x, hd = dim >> 1;
    const float t = cur[b] / tot[b] * scale;
    for (int i = threadIdx.x; i < hd; i += blockDim.x) {
        const float f = expf(-logf(10000.0f) * (float)i / (float)hd);
        te[(int64_t)b * dim + i] = sinf(t * f);
        te[(int64_t)b * dim + hd + i] = cosf(t * f);
    }
}
void launch_time_embed(hipStream_t s, const float* cur, const float* tot, int B, int dim, float scale, float* te) {
    if (B == 0) return;
    STN_KLAUNCH(time_embed_kernel, dim3(B), dim3(64), 0, s, cur, tot, dim, scale, te);
}

// ---------------------------------------------------------------------------------------------
// vocoder front: frame (b, t = l*ccf + j) channel c  <-  latent[b][j*ld + c][l];   conv ld -> C, kernel k
// 8 frames per workgroup; the (8 + k - 1) x ld input window sits in LDS; weights [ld*k][C] (co contiguous).
// ---------------------------------------------------------------------------------------------
static constexpr int VI_FR = 8;
__global__ __launch_bounds__(256) void vocoder_in_kernel(const float* __restrict__ latent, int L, int ld, int ccf,
                                                         const float* __restrict__ w_t, const float* __restrict__ bias,
                                                         int C, int k, float* __restrict__ x, const int* __restrict__ seqlen) {
    extern __shared__ __attribute__((aligned(16))) float win[];  // [(VI_FR + k - 1)][ld]
    const int T = L * ccf, D = ld * ccf;
    const int tiles = (T + VI_FR - 1) / VI_FR;
    const int b = blockIdx.x / tiles, t0 = (blockIdx.x % tiles) * VI_FR;
    const int half = (k - 1) >> 1, nwin = VI_FR + k - 1;
    for (int i = threadIdx.x; i < nwin * ld; i += blockDim.x) {
        const int f = i / ld, c = i - f * ld;
        const int t = t0 - half + f;
        float v = 0.f;
        if (t >= 0 && t < (seqlen ? seqlen[b] : T)) {
            const int l = t / ccf, j = t - l * ccf;
            v = latent[((int64_t)b * D + j * ld + c) * L + l];
        }
        win[i] = v;
    }
    __syncthreads();
    for (int co = threadIdx.x; co < C; co += blockDim.x) {
        float acc[VI_FR];
        const float bv = bias[co];
#pragma unroll
        for (int f = 0; f < VI_FR; ++f) acc[f] = bv;
        for (int ci = 0; ci < ld; ++ci)
            for (int j = 0; j < k; ++j) {
                const float wv = w_t[(int64_t)(ci * k + j) * C + co];
#pragma unroll
                for (int f = 0; f < VI_FR; ++f) acc[f] = fmaf(wv, win[(f + j) * ld + ci], acc[f]);
            }
#pragma unroll
        for (int f = 0; f < VI_FR; ++f)
            if (t0 + f < T) x[((int64_t)b * T + t0 + f) * C + co] = acc[f];
    }
}
void launch_vocoder_in(hipStream_t s, const float* latent, int B, int L, int ld, int ccf, const float* w_t,
                       const float* bias, int C, int k, float* x, const int* seqlen) {
    const int T = L * ccf;
    if (B * T == 0) return;
    const int tiles = (T + VI_FR - 1) / VI_FR;
    const size_t lds = sizeof(float) * (size_t)(VI_FR + k - 1) * ld;
    STN_KLAUNCH(vocoder_in_kernel, dim3(B * tiles), dim3(256), lds, s, latent, L, ld, ccf, w_t, bias, C, k, x, seqlen);
}

// cols[(b, t)][ci * k + j] = latent frame (t + j - (k-1)/2) of sequence b, channel ci (zero outside the sequence; columns >= ld * k are the GEMM's K
// padding), where frame tt of the vocoder is latent position l = tt / ccf, channel block q = tt % ccf: latent[b][q * ld + ci][l].
// A workgroup owns IM2_TF consecutive frames of one sequence: it stages the latent positions they touch in LDS ([channel][l], the reads run along l) and
// writes whole rows of cols, consecutive lanes consecutive columns (the former one-thread-per-element form gathered 4 bytes per lane with four integer
// divisions each: 38 us for the bench's 23 MB).
constexpr int IM2_TF = 32;
// FIXED: the published shape (24 latent channels x 6, k = 7, K padded to 192) as compile-time constants — the index arithmetic is three divisions per
// element, and with run-time divisors they, not the bytes, set the kernel's time; with them and eight columns per 16-byte store 38 -> 17 us
template <typename OutT, bool FIXED>
__global__ __launch_bounds__(256) void vocoder_im2col_kernel(const float* __restrict__ latent, int L, int ld_, int ccf_, int k_, int kp_,
                                                             OutT* __restrict__ cols, const int* __restrict__ seqlen, const int* __restrict__ row_off) {
    extern __shared__ float im2_sm[];  // [D][NL]
    const int ld = FIXED ? 24 : ld_, ccf = FIXED ? 6 : ccf_, k = FIXED ? 7 : k_, kp = FIXED ? 192 : kp_;
    const int T = L * ccf, D = ld * ccf, half = (k - 1) >> 1;
    const int tiles = (T + IM2_TF - 1) / IM2_TF;
    const int b = (int)blockIdx.x / tiles, t0 = ((int)blockIdx.x % tiles) * IM2_TF;
    const int n = seqlen ? seqlen[b] : T;            // frames of this sequence that exist (taps beyond read zero)
    if (row_off && t0 >= n) return;                  // packed destination: no such rows
    const int l0 = max(t0 - half, 0) / ccf, l1 = min((min(t0 + IM2_TF, T) - 1 + half) / ccf, L - 1), NL = l1 - l0 + 1;
    for (int i = threadIdx.x; i < D * NL; i += 256) {
        const int d = i / NL, l = i - d * NL;
        im2_sm[i] = latent[((int64_t)b * D + d) * L + l0 + l];
    }
    __syncthreads();
    const int t1 = min(t0 + IM2_TF, row_off ? n : T);
    const int64_t row00 = row_off ? (int64_t)row_off[b] : (int64_t)b * T;
    auto value = [&](int t, int col) {
        float v = 0.f;
        if (col < ld * k) {
            const int ci = col / k, j = col - ci * k, tt = t + j - half;
            if (tt >= 0 && tt < n) { const int l = tt / ccf, q = tt - l * ccf; v = im2_sm[(q * ld + ci) * NL + (l - l0)]; }
        }
        return v;
    };
    if constexpr (FIXED && sizeof(OutT) == 2) {  // eight columns per thread: one 16-byte store (kp = 192 = 24 x 8)
        for (int i = threadIdx.x; i < (t1 - t0) * 24; i += 256) {
            const int r = i / 24, c8 = i - r * 24, t = t0 + r;
            OutT tmp[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) store1(tmp + e, value(t, c8 * 8 + e));
            *reinterpret_cast<uint4*>(cols + (row00 + t) * kp + c8 * 8) = *reinterpret_cast<const uint4*>(tmp);
        }
    } else {
        for (int i = threadIdx.x; i < (t1 - t0) * kp; i += 256) {
            const int r = i / kp, col = i - r * kp, t = t0 + r;
            store1(cols + (row00 + t) * kp + col, value(t, col));
        }
    }
}
void launch_vocoder_im2col(hipStream_t s, int out_dtype, const float* latent, int B, int L, int ld, int ccf, int k, int kp, void* cols,
                           const int* seqlen, const int* row_off) {
    if ((int64_t)B * L * ccf * kp == 0) return;
    const int T = L * ccf, tiles = (T + IM2_TF - 1) / IM2_TF;
    const int nl_max = (IM2_TF + k - 1) / ccf + 2;
    const size_t lds = sizeof(float) * (size_t)ld * ccf * nl_max;
    if (lds > 64 * 1024 || (int64_t)B * tiles > 0x7FFFFFFFll) throw std::invalid_argument("launch_vocoder_im2col: latent window does not fit the staging buffer");
    const dim3 grid((unsigned)(B * tiles));
    const bool fixed = ld == 24 && ccf == 6 && k == 7 && kp == 192;
#define STN_IM2(T_) do { if (fixed) STN_KLAUNCH((vocoder_im2col_kernel<T_, true>), grid, dim3(256), lds, s, latent, L, ld, ccf, k, kp, static_cast<T_*>(cols), seqlen, row_off); \
                         else STN_KLAUNCH((vocoder_im2col_kernel<T_, false>), grid, dim3(256), lds, s, latent, L, ld, ccf, k, kp, static_cast<T_*>(cols), seqlen, row_off); } while (0)
    if (out_dtype == F16) STN_IM2(f16_t);
    else if (out_dtype == BF16) STN_IM2(uint16_t);
    else STN_IM2(float);
#undef STN_IM2
}

template <typename InT>
__global__ void masked_mean_kernel(const InT* __restrict__ x, int L, int C, const int* __restrict__ len,
                                   float* __restrict__ pooled, const int* __restrict__ row_off) {
    const int b = blockIdx.x, n = len[b];
    const int64_t r0 = row_off ? (int64_t)row_off[b] : (int64_t)b * L;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int t = 0; t < n; ++t) s += load1(x + (r0 + t) * C + c);
        pooled[(int64_t)b * C + c] = s / (float)(n > 0 ? n : 1);
    }
}
void launch_masked_mean(hipStream_t s, int in_dtype, const void* x, int B, int L, int C, const int* len, float* pooled,
                        const int* row_off) {
    if (B == 0) return;
    if (in_dtype == F16) STN_KLAUNCH(masked_mean_kernel<f16_t>, dim3(B), dim3(128), 0, s, static_cast<const f16_t*>(x), L, C, len, pooled, row_off);
    else if (in_dtype == BF16) STN_KLAUNCH(masked_mean_kernel<uint16_t>, dim3(B), dim3(128), 0, s, static_cast<const uint16_t*>(x), L, C, len, pooled, row_off);
    else STN_KLAUNCH(masked_mean_kernel<float>, dim3(B), dim3(128), 0, s, static_cast<const float*>(x), L, C, len, pooled, row_off);
}

__global__ void softplus_kernel(float* x, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float y = x[i]; x[i] = y > 20.f ? y : log1pf(expf(y)); }
}
void launch_softplus(hipStream_t s, float* x, int n) {
    if (n) STN_KLAUNCH(softplus_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, n);
}
__global__ void scale_kernel(float* x, int n, float mul) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= mul;
}
void launch_scale(hipStream_t s, float* x, int n, float mul) {
    if (n) STN_KLAUNCH(scale_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, n, mul);
}
__global__ void reciprocal_kernel(const float* in, int n, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = 1.0f / in[i];
}
void launch_reciprocal(hipStream_t s, const float* in, int n, float* out) {
    if (n) STN_KLAUNCH(reciprocal_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in, n, out);
}
template <typename InT>
__global__ void half_to_f32_kernel(const InT* __restrict__ in, int64_t n, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = load1(in + i);
}
void launch_half_to_f32(hipStream_t s, int in_dtype, const void* in, int64_t n, float* out) {
    if (!n) return;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (in_dtype == F16) STN_KLAUNCH(half_to_f32_kernel<f16_t>, grid, dim3(256), 0, s, static_cast<const f16_t*>(in), n, out);
    else STN_KLAUNCH(half_to_f32_kernel<uint16_t>, grid, dim3(256), 0, s, static_cast<const uint16_t*>(in), n, out);
}
void launch_bf16_to_f32(hipStream_t s, const uint16_t* in, int64_t n, float* out) { launch_half_to_f32(s, BF16, in, n, out); }
__global__ void fill_kernel(float* x, int n, float v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = v;
}
void launch_fill(hipStream_t s, float* x, int n, float v) {
    if (n) STN_KLAUNCH(fill_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, n, v);
}
// the step counters of a whole Euler loop in one launch: tot[st][b] = steps, cur[st][b] = st, dt[b] = 1 / steps
__global__ void step_counters_kernel(float* __restrict__ tot, float* __restrict__ cur, float* __restrict__ dt, int B, int steps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * steps) { tot[i] = (float)steps; cur[i] = (float)(i / B); }
    if (i < B) dt[i] = 1.0f / (float)steps;
}
void launch_step_counters(hipStream_t s, float* tot, float* cur, float* dt, int B, int steps) {
    const int n = B * steps;
    if (n) STN_KLAUNCH(step_counters_kernel, dim3((n + 255) / 256), dim3(256), 0, s, tot, cur, dt, B, steps);
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller; element (utt, d, t) depends only on (seed, utt, d, t)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__global__ void randn_masked_kernel(unsigned long long seed, const unsigned long long* __restrict__ seed_dev,
                                    const int64_t* __restrict__ utt_ids, int D, int L, const int* __restrict__ len, int64_t n4,
                                    float* __restrict__ xt) {
    if (seed_dev) seed = *seed_dev;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over [B][D][ceil(L/4)]
    if (i >= n4) return;
    const int L4 = (L + 3) >> 2;
    const int t4 = (int)(i % L4);
    const int64_t r = i / L4;
    const int d = (int)(r % D);
    const int b = (int)(r / D);
    const unsigned long long u = utt_ids ? (unsigned long long)utt_ids[b] : (unsigned long long)b;
    unsigned c[4] = {(unsigned)t4, (unsigned)d, (unsigned)u, (unsigned)(u >> 32)};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
    float nrm[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float u2 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float rr = sqrtf(-2.0f * logf(u1)), th = 6.28318530717958647692f * u2;
        nrm[2 * h] = rr * cosf(th);
        nrm[2 * h + 1] = rr * sinf(th);
    }
    const int nb = len ? len[b] : L;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int t = t4 * 4 + q;
        if (t < L) xt[((int64_t)b * D + d) * L + t] = t < nb ? nrm[q] : 0.f;
    }
}
void launch_randn_masked(hipStream_t s, uint64_t seed, const int64_t* utt_ids, int B, int D, int L, const int* len,
                         float* xt, const unsigned long long* seed_dev) {
    const int64_t n4 = (int64_t)B * D * ((L + 3) / 4);
    if (n4 == 0) return;
    STN_KLAUNCH(randn_masked_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (unsigned long long)seed,
                       seed_dev, utt_ids, D, L, len, n4, xt);
}

__global__ void scale_len_kernel(const int* __restrict__ len, int B, int factor, int* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) out[i] = len[i] * factor;
}
void launch_scale_len(hipStream_t s, const int* len, int B, int factor, int* out) {
    if (B == 0) return;
    STN_KLAUNCH(scale_len_kernel, dim3((B + 255) / 256), dim3(256), 0, s, len, B, factor, out);
}

// Extents of the exact "trimmed" dense vocoder: an utterance whose zero-latent padding is longer than twice the receptive
// field rf is computed on len*ccf + 2*rf frames (zero beyond), which is exact on its first len*ccf + rf output frames; the rest
// of its row is position-independent (quiet chunk + edge tail, see Engine::prepare_vocoder_constants).
__global__ void trim_len_kernel(const int* __restrict__ len, int B, int ccf, int T, int rf, int* __restrict__ n_out,
                                int* __restrict__ valid_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const int l6 = len[i] * ccf;
    const bool trim = l6 + 2 * rf <= T;  // the edge tail's whole receptive field [T - 2rf, T) is zero latent (quiet part may be empty)
    n_out[i] = trim ? l6 + 2 * rf : T;
    valid_out[i] = trim ? l6 + rf : T;
}
void launch_trim_len(hipStream_t s, const int* len, int B, int ccf, int T, int rf, int* n_out, int* valid_out) {
    if (B == 0) return;
    STN_KLAUNCH(trim_len_kernel, dim3((B + 255) / 256), dim3(256), 0, s, len, B, ccf, T, rf, n_out, valid_out);
}

// packed rows -> padded [B][T][W]: computed frames below valid[b], then the quiet chunk, then the edge tail of rf frames
__global__ void unpack_rows_quiet_kernel(const float* __restrict__ src, const int* __restrict__ valid, const int* __restrict__ row_off,
                                         int T, int W4, int rf, const float* __restrict__ quiet, const float* __restrict__ edge,
                                         int64_t n4, float* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over [B][T][W/4]
    if (i >= n4) return;
    const int c = (int)(i % W4);
    const int64_t r = i / W4;
    const int t = (int)(r % T);
    const int b = (int)(r / T);
    float4 v;
    if (t < valid[b]) v = reinterpret_cast<const float4*>(src)[((int64_t)row_off[b] + t) * W4 + c];
    else if (t >= T - rf) v = reinterpret_cast<const float4*>(edge)[(int64_t)(t - (T - rf)) * W4 + c];
    else v = reinterpret_cast<const float4*>(quiet)[c];
    reinterpret_cast<float4*>(dst)[i] = v;
}
void launch_unpack_rows_quiet(hipStream_t s, const float* src, const int* valid, const int* row_off, int B, int T, int W, int rf,
                              const float* quiet, const float* edge, float* dst) {
    const int64_t n4 = (int64_t)B * T * (W / 4);
    if (n4 == 0) return;
    if (W % 4) { throw std::invalid_argument("unpack_rows needs W % 4 == 0"); }
    STN_KLAUNCH(unpack_rows_quiet_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, src, valid, row_off, T, W / 4, rf, quiet, edge,
                n4, dst);
}

__global__ void mask_ncl_kernel(float* __restrict__ x, int D, int L, int64_t n, const int* __restrict__ len) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = (int)(i % L);
    const int b = (int)(i / ((int64_t)D * L));
    if (t >= len[b]) x[i] = 0.f;
}
void launch_mask_ncl(hipStream_t s, float* x, int B, int D, int L, const int* len) {
    const int64_t n = (int64_t)B * D * L;
    if (n == 0) return;
    STN_KLAUNCH(mask_ncl_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, D, L, n, len);
}

// fp32 -> int16 PCM exactly as the reference's writeWavFile (clamp to [-1,1], * 32767, truncation toward zero), 8 samples
// per thread (two 16-B loads, one 16-B store); rows of W samples may land with a destination stride (gather payloads).
__global__ void pcm16_kernel(const float* __restrict__ w, int W8, int64_t n8, int16_t* __restrict__ pcm, int64_t dst_stride) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over [rows][W/8]
    if (i >= n8) return;
    const int64_t row = i / W8;
    const int c = (int)(i - row * W8);
    const float4* src = reinterpret_cast<const float4*>(w) + i * 2;
    const float4 a = src[0], b = src[1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int lo = (int)(fminf(1.0f, fmaxf(-1.0f, v[2 * j])) * 32767.0f);
        const int hi = (int)(fminf(1.0f, fmaxf(-1.0f, v[2 * j + 1])) * 32767.0f);
        o[j] = ((unsigned)lo & 0xFFFFu) | ((unsigned)hi << 16);
    }
    *reinterpret_cast<uint4*>(pcm + row * dst_stride + (int64_t)c * 8) = make_uint4(o[0], o[1], o[2], o[3]);
}
__global__ void pcm16_scalar_kernel(const float* __restrict__ w, int W, int64_t n, int16_t* __restrict__ pcm, int64_t dst_stride) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t row = i / W;
    const float c = fminf(1.0f, fmaxf(-1.0f, w[i]));
    pcm[row * dst_stride + (i - row * W)] = (int16_t)(int)(c * 32767.0f);
}
void launch_f32_to_pcm16(hipStream_t s, const float* w, int64_t rows, int W, int16_t* pcm, int64_t dst_stride) {
    const int64_t n = rows * W;
    if (n == 0) return;
    if (W % 8 == 0 && dst_stride % 8 == 0 && !(reinterpret_cast<uintptr_t>(pcm) & 15) && !(reinterpret_cast<uintptr_t>(w) & 15)) {
        const int64_t n8 = n / 8;
        STN_KLAUNCH(pcm16_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, w, W / 8, n8, pcm, dst_stride);
    } else {
        STN_KLAUNCH(pcm16_scalar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, W, n, pcm, dst_stride);
    }
}

}  // namespace stn
