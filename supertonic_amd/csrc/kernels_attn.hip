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

#include <stdio.h>
#include <stdlib.h>

namespace stn {

static constexpr int AQ = 32, AK = 64, ADH_MAX = 96;

__device__ __forceinline__ float ld_act(const float* p) { return *p; }
__device__ __forceinline__ float ld_act(const uint16_t* p) { return __uint_as_float(((unsigned)*p) << 16); }
__device__ __forceinline__ void st_act(float* p, float v) { *p = v; }
__device__ __forceinline__ void st_act(uint16_t* p, float v) {
    // round-to-nearest-even; inputs are finite softmax averages
    const unsigned u = __float_as_uint(v);
    *p = (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// stage `rows` rows (starting at sequence position pos0 of batch b) of one head into LDS, rotating pairs
// (i, i + dh/2) when rope_mode >= 0; rows at positions >= limit are zero-filled.
template <typename T>
__device__ __forceinline__ void stage_rows(const T* __restrict__ src, int ld, int64_t seq_base, int pos0, int rows,
                                           int limit, int dh, int ds, float* __restrict__ dst, int rope_mode,
                                           float log_base, float gamma, int seq_len, float mul) {
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

template <typename T>
__global__ __launch_bounds__(256) void attn_kernel(const T* __restrict__ q, int ldq, const T* __restrict__ k,
                                                   const T* __restrict__ v, int ldk, T* __restrict__ o, int ldo, int Lq,
                                                   int Lk, int dh, const int* __restrict__ qlen,
                                                   const int* __restrict__ klen, int rope_mode, float log_base,
                                                   float gamma) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ds = dh + 1;
    float* Qs = lds;               // [AQ][ds]
    float* Ks = Qs + AQ * ds;      // [AK][ds]
    float* Vs = Ks + AK * ds;      // [AK][ds]
    float* Ss = Vs + AK * ds;      // [AQ][AK + 1]
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * AQ;
    const int tid = threadIdx.x, qi = tid >> 3, g = tid & 7;
    const int nk = klen ? min(klen[b], Lk) : Lk;
    const int nq = qlen ? qlen[b] : Lq;  // used for length-aware positions only
    const float sc = rsqrtf((float)dh);

    stage_rows<T>(q + h * dh, ldq, (int64_t)b * Lq, q0, AQ, Lq, dh, ds, Qs, rope_mode, log_base, gamma, nq, sc);

    float m_run = -1e30f, l_run = 0.f;
    float oacc[ADH_MAX / 8];
#pragma unroll
    for (int i = 0; i < ADH_MAX / 8; ++i) oacc[i] = 0.f;

    for (int k0 = 0; k0 < nk; k0 += AK) {
        __syncthreads();  // previous tile fully consumed (and Qs visible on the first pass)
        stage_rows<T>(k + h * dh, ldk, (int64_t)b * Lk, k0, AK, nk, dh, ds, Ks, rope_mode, log_base, gamma, nk, 1.f);
        stage_rows<T>(v + h * dh, ldk, (int64_t)b * Lk, k0, AK, nk, dh, ds, Vs, -1, 0.f, 0.f, 1, 1.f);
        __syncthreads();
        float s[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) s[jj] = 0.f;
        const float* qrow = Qs + qi * ds;
        for (int d = 0; d < dh; ++d) {
            const float qv = qrow[d];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) s[jj] = fmaf(qv, Ks[(g + 8 * jj) * ds + d], s[jj]);
        }
        float mx = -1e30f;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            if (k0 + g + 8 * jj >= nk) s[jj] = -1e30f;
            mx = fmaxf(mx, s[jj]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = expf(m_run - m_new);
        float ps = 0.f;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const float p = s[jj] <= -1e29f ? 0.f : expf(s[jj] - m_new);
            ps += p;
            Ss[qi * (AK + 1) + g + 8 * jj] = p;
        }
        ps += __shfl_xor(ps, 1, 64);
        ps += __shfl_xor(ps, 2, 64);
        ps += __shfl_xor(ps, 4, 64);
        l_run = l_run * alpha + ps;
        m_run = m_new;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ADH_MAX / 8; ++i) oacc[i] *= alpha;
        const float* prow = Ss + qi * (AK + 1);
        for (int j = 0; j < AK; ++j) {
            const float p = prow[j];
            const float* vrow = Vs + j * ds + g;
#pragma unroll
            for (int i = 0; i < ADH_MAX / 8; ++i)
                if (8 * i < dh) oacc[i] = fmaf(p, vrow[8 * i], oacc[i]);
        }
    }
    const int gq = q0 + qi;
    if (gq < Lq) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        T* orow = o + ((int64_t)b * Lq + gq) * ldo + h * dh + g;
#pragma unroll
        for (int i = 0; i < ADH_MAX / 8; ++i)
            if (8 * i + g < dh) st_act(orow + 8 * i, oacc[i] * inv);
    }
}

void launch_attention(hipStream_t s, int dtype, const void* q, int ldq, const void* k, const void* v, int ldk, void* o,
                      int ldo, int B, int Lq, int Lk, int H, int dh, const int* qlen, const int* klen, int rope_mode,
                      float rope_base, float rope_gamma) {
    if (B == 0 || Lq == 0) return;
    if (dh > ADH_MAX || dh % 8 || dh < 8) { fprintf(stderr, "stn: attention head dim %d unsupported (multiple of 8, <= %d)\n", dh, ADH_MAX); abort(); }
    const int ds = dh + 1;
    const size_t lds = sizeof(float) * ((size_t)(AQ + 2 * AK) * ds + (size_t)AQ * (AK + 1));
    const dim3 grid((Lq + AQ - 1) / AQ, H, B);
    const float log_base = logf(rope_base);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<uint16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_set = true;
    }
    if (dtype == BF16)
        hipLaunchKernelGGL(attn_kernel<uint16_t>, grid, dim3(256), lds, s, static_cast<const uint16_t*>(q), ldq,
                           static_cast<const uint16_t*>(k), static_cast<const uint16_t*>(v), ldk, static_cast<uint16_t*>(o), ldo,
                           Lq, Lk, dh, qlen, klen, rope_mode, log_base, rope_gamma);
    else
        hipLaunchKernelGGL(attn_kernel<float>, grid, dim3(256), lds, s, static_cast<const float*>(q), ldq,
                           static_cast<const float*>(k), static_cast<const float*>(v), ldk, static_cast<float*>(o), ldo, Lq, Lk,
                           dh, qlen, klen, rope_mode, log_base, rope_gamma);
}

}  // namespace stn
