// kernels_gemm.hip — MFMA GEMM with fused epilogues for gfx950 (MI355X).
//
//   acc[m][n] = sum_k A[m][k] * W[n][k]        A: activations [M][K], W: weights [N][K] (both K-contiguous)
//
// Used for every dense linear / pointwise conv of the four graphs that replace the reference's
// Ort::Session::Run sites (/root/reference/cpp/helper.cpp:519,552,643,668).
//
// Tiling (wave64, 4 waves = 2x2, each wave a 64x64 output sub-tile = 2x2 MFMA 32x32 tiles):
//   block tile 128 x 128, K-step = 128 bytes of K per row (64 bf16 / 32 f32)
//   LDS: 2 stages x (A 16 KiB + W 16 KiB) = 64 KiB  -> 2 workgroups per CU
//   global -> registers (16-B loads, issued before the MFMAs of the current stage) -> LDS (after them):
//   one barrier per K-step.
//   bf16: v_mfma_f32_32x32x16_bf16, fragments by ds_read_b128 from an XOR-swizzled image
//         (16-B slot = chunk ^ ((row >> 1) & 7): conflict-free for the b128 16-lane groups)
//   f32 : v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain), fragments by ds_read_b32 from a
//         word-swizzled image (word = k ^ (row & 31): conflict-free per 32-lane half)
// Workgroup -> tile map is XCD-aware: the tiles that share an A row-panel are consecutive and land on
// one XCD (private 4 MiB L2), the bijective remap of the CDNA4 guide.
#include "kernels.hpp"

#include <hip/hip_bf16.h>
#include <stdio.h>
#include <stdlib.h>

namespace stn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// Out-of-range lanes get this byte offset: beyond num_records, so the buffer load returns zeros
// (hardware range check) — no divergent branch, no select-of-pointers.
static constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, size_t bytes) {
    const unsigned n = bytes > 0x7FFFFFFFu ? 0x7FFFFFFFu : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, n, 0x00020000);
}

static constexpr int BM = 128, BN = 128, NT = 256;
static constexpr int STAGE_BYTES = (BM + BN) * 128;  // 32 KiB

// erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7): GELU(x) = 0.5 x (1 + erf(x / sqrt2))
__device__ __forceinline__ float gelu_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __frcp_rn(1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * __expf(-z * z);
    const float erf_v = x < 0.f ? -erf_abs : erf_abs;
    return 0.5f * x * (1.0f + erf_v);
}
__device__ __forceinline__ float act_f(float v, int act) {
    if (act == ACT_GELU) return gelu_f(v);
    if (act == ACT_SILU) return v / (1.0f + expf(-v));
    return v;
}
__device__ __forceinline__ uint16_t f2bf(float f) {
    __hip_bfloat16 h = __float2bfloat16(f);
    return *reinterpret_cast<uint16_t*>(&h);
}

// XCD-aware bijective remap of a 1-D grid (consecutive logical tiles -> same XCD).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + loc;
}

// One lane's share of the epilogue: accumulator tile (mi, ni) element i lives at
//   row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5), col = lane & 31      (32x32 MFMA C/D map)
template <int MODE>
__device__ __forceinline__ void run_epilogue(const Epilogue& e, f32x16 (&acc)[2][2], int m_base, int n_base, int M,
                                             int N, int lane) {
    const int half = lane >> 5, cl = lane & 31;
    float bias[2], gam[2];
    int ncol[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        ncol[ni] = n_base + ni * 32 + cl;
        const bool ok = ncol[ni] < N;
        bias[ni] = (ok && e.bias) ? e.bias[ncol[ni]] : 0.f;
        gam[ni] = (ok && e.gamma) ? e.gamma[ncol[ni]] : 1.f;
    }
    const bool need_bt = (e.len != nullptr) || MODE >= EPI_EULER_T;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = m_base + mi * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
            if (m >= M) continue;
            int b = 0, t = m;
            float keep = 1.f;
            if (need_bt) {
                b = m / e.L;
                t = m - b * e.L;
                if (e.len && t >= e.len[b]) keep = 0.f;
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int n = ncol[ni];
                if (n >= N) continue;
                const float v = acc[mi][ni][i] + bias[ni];
                if (MODE == EPI_STORE) {
                    const float r = act_f(v, e.act) * keep;
                    const size_t o = (size_t)m * e.ldo + n;
                    if (e.out_dtype == BF16) reinterpret_cast<uint16_t*>(e.out)[o] = f2bf(r);
                    else reinterpret_cast<float*>(e.out)[o] = r;
                } else if (MODE == EPI_RESID) {
                    const size_t o = (size_t)m * e.ldo + n;
                    e.resid[o] = (e.resid[o] + gam[ni] * v) * keep;
                } else if (MODE == EPI_EULER_T) {
                    const size_t o = ((size_t)b * N + n) * e.L + t;
                    reinterpret_cast<float*>(e.out)[o] = keep != 0.f ? (e.aux[o] + v * e.row_scale[b]) : 0.f;
                } else {  // EPI_STORE_T
                    const size_t o = ((size_t)b * N + n) * e.L + t;
                    reinterpret_cast<float*>(e.out)[o] = v * keep;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// bf16 operands, fp32 accumulate
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(NT, 2) void gemm_bf16_kernel(const uint16_t* __restrict__ A, int lda,
                                                          const uint16_t* __restrict__ W, int ldw, int M, int N, int K,
                                                          int tiles_n, int ntiles, Epilogue e) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = xcd_remap(blockIdx.x, ntiles);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;

    const int sc = tid & 7, sr = tid >> 3;  // staging: 16-B chunk along K, row (4 rows per thread, 32 apart)
    u32x4 ra[4], rb[4];
    // per-block descriptors: rows m0.. of A and n0.. of W; a row past the end is out of range by construction
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A + (size_t)m0 * lda, m0 < M ? (size_t)(M - m0) * lda * 2 : 0);
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(W + (size_t)n0 * ldw, n0 < N ? (size_t)(N - n0) * ldw * 2 : 0);

#define STN_GLOAD(k0)                                                                              \
    {                                                                                              \
        const int gk = (k0) + sc * 8;                                                              \
        const bool kok = gk < K;                                                                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
            const int row = sr + 32 * i;                                                           \
            const unsigned oa = kok ? (unsigned)(row * lda + gk) * 2u : OOB;                       \
            const unsigned ow = kok ? (unsigned)(row * ldw + gk) * 2u : OOB;                       \
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, oa, 0, 0);                          \
            rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsW, ow, 0, 0);                          \
        }                                                                                          \
    }
#define STN_SWRITE(buf)                                                          \
    {                                                                            \
        unsigned char* sa_ = smem + (buf) * STAGE_BYTES;                         \
        unsigned char* sb_ = sa_ + BM * 128;                                     \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                          \
            const int row = sr + 32 * i;                                         \
            const int off = row * 128 + ((sc ^ ((row >> 1) & 7)) << 4);          \
            *reinterpret_cast<u32x4*>(sa_ + off) = ra[i];                        \
            *reinterpret_cast<u32x4*>(sb_ + off) = rb[i];                        \
        }                                                                        \
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

    const int nk = (K + 63) >> 6;
    STN_GLOAD(0);
    STN_SWRITE(0);
    __syncthreads();

    const int lr = lane & 31, lh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) STN_GLOAD((kt + 1) << 6);
        const unsigned char* sa = smem + (kt & 1) * STAGE_BYTES;
        const unsigned char* sb = sa + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int chunk = ks * 2 + lh;
            bf16x8 a[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int row = wm * 64 + mi * 32 + lr;
                a[mi] = *reinterpret_cast<const bf16x8*>(sa + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int row = wn * 64 + ni * 32 + lr;
                b[ni] = *reinterpret_cast<const bf16x8*>(sb + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        if (more) STN_SWRITE((kt + 1) & 1);
        __syncthreads();
    }
#undef STN_GLOAD
#undef STN_SWRITE
    run_epilogue<MODE>(e, acc, m0 + wm * 64, n0 + wn * 64, M, N, lane);
}

// ---------------------------------------------------------------------------------------------
// fp32 operands, exact fp32 MFMA
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(NT, 2) void gemm_f32_kernel(const float* __restrict__ A, int lda,
                                                         const float* __restrict__ W, int ldw, int M, int N, int K,
                                                         int tiles_n, int ntiles, Epilogue e) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = xcd_remap(blockIdx.x, ntiles);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;

    const int sc = tid & 7, sr = tid >> 3;
    u32x4 ra[4], rb[4];
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A + (size_t)m0 * lda, m0 < M ? (size_t)(M - m0) * lda * 4 : 0);
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(W + (size_t)n0 * ldw, n0 < N ? (size_t)(N - n0) * ldw * 4 : 0);

#define STN_GLOAD(k0)                                                                              \
    {                                                                                              \
        const int gk = (k0) + sc * 4;                                                              \
        const bool kok = gk < K;                                                                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
            const int row = sr + 32 * i;                                                           \
            const unsigned oa = kok ? (unsigned)(row * lda + gk) * 4u : OOB;                       \
            const unsigned ow = kok ? (unsigned)(row * ldw + gk) * 4u : OOB;                       \
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, oa, 0, 0);                          \
            rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsW, ow, 0, 0);                          \
        }                                                                                          \
    }
#define STN_SWRITE(buf)                                                          \
    {                                                                            \
        unsigned* sa_ = reinterpret_cast<unsigned*>(smem + (buf) * STAGE_BYTES);       \
        unsigned* sb_ = sa_ + BM * 32;                                            \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                          \
            const int row = sr + 32 * i;                                         \
            const int sw = row & 31, kb = sc * 4;                                \
            sa_[row * 32 + ((kb + 0) ^ sw)] = ra[i].x;                           \
            sa_[row * 32 + ((kb + 1) ^ sw)] = ra[i].y;                           \
            sa_[row * 32 + ((kb + 2) ^ sw)] = ra[i].z;                           \
            sa_[row * 32 + ((kb + 3) ^ sw)] = ra[i].w;                           \
            sb_[row * 32 + ((kb + 0) ^ sw)] = rb[i].x;                           \
            sb_[row * 32 + ((kb + 1) ^ sw)] = rb[i].y;                           \
            sb_[row * 32 + ((kb + 2) ^ sw)] = rb[i].z;                           \
            sb_[row * 32 + ((kb + 3) ^ sw)] = rb[i].w;                           \
        }                                                                        \
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

    const int nk = (K + 31) >> 5;
    STN_GLOAD(0);
    STN_SWRITE(0);
    __syncthreads();

    const int lr = lane & 31, lh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) STN_GLOAD((kt + 1) << 5);
        const float* sa = reinterpret_cast<const float*>(smem + (kt & 1) * STAGE_BYTES);
        const float* sb = sa + BM * 32;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int k = ks * 2 + lh;
            float a[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int row = wm * 64 + mi * 32 + lr;
                a[mi] = sa[row * 32 + (k ^ (row & 31))];
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int row = wn * 64 + ni * 32 + lr;
                b[ni] = sb[row * 32 + (k ^ (row & 31))];
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        if (more) STN_SWRITE((kt + 1) & 1);
        __syncthreads();
    }
#undef STN_GLOAD
#undef STN_SWRITE
    run_epilogue<MODE>(e, acc, m0 + wm * 64, n0 + wn * 64, M, N, lane);
}


// ---------------------------------------------------------------------------------------------
// bf16 v2: LDS-DMA ring.  K % 32 == 0.
//   K-step = 32 bf16 (64-B rows): stage = A 8 KiB + W 8 KiB; 4 stages = 64 KiB -> 2 workgroups per CU.
//   Operands go HBM/L2 -> LDS directly (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction, hardware
//   range check zero-fills rows past M / N); three stages stay in flight across raw s_barriers behind a
//   COUNTED s_waitcnt vmcnt, so the per-CU ingest pipe never drains while the MFMAs run.
//   LDS image: 16-B slot = chunk ^ ((row >> 2) & 3), applied on the per-lane SOURCE address (the DMA writes
//   lane-linear) and again on the ds_read_b128 fragment reads: conflict-free 16-lane groups.
//   Epilogue (STORE / RESID): accumulators -> LDS as fp32 [128][128] (the ring is dead by then) -> each lane
//   owns 8 consecutive columns of a row: bias/GELU/residual on 16-B vectors, 16-B coalesced global stores.
// ---------------------------------------------------------------------------------------------
static constexpr int RK = 32, RSTAGES = 4, RSTAGE_BYTES = (BM + BN) * RK * 2;  // 16 KiB

#define STN_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int MODE, bool VEC>
__global__ __launch_bounds__(NT, 2) void gemm_bf16_ring_kernel(const uint16_t* __restrict__ A, int lda,
                                                               const uint16_t* __restrict__ W, int ldw, int M, int N,
                                                               int K, int tiles_n, int ntiles, Epilogue e) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[RSTAGES * RSTAGE_BYTES];  // 64 KiB, the only LDS object
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = xcd_remap(blockIdx.x, ntiles);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;

    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A + (size_t)m0 * lda, m0 < M ? (size_t)(M - m0) * lda * 2 : 0);
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(W + (size_t)n0 * ldw, n0 < N ? (size_t)(N - n0) * ldw * 2 : 0);

    // this wave's DMA pieces: A pieces 2w, 2w+1 and W pieces 2w, 2w+1 (a piece = 16 rows x 64 B = 1 KiB)
    unsigned offA[2], offW[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (2 * wave + j) * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);  // source-side swizzle
        offA[j] = (unsigned)(row * lda + chunk * 8) * 2u;
        offW[j] = (unsigned)(row * ldw + chunk * 8) * 2u;
    }
    const int nk = K / RK;

#define STN_ISSUE(kt)                                                                                              \
    {                                                                                                              \
        unsigned char* st_ = smem + ((kt) & (RSTAGES - 1)) * RSTAGE_BYTES;                                         \
        const int ko_ = (kt) * (RK * 2);                                                                           \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, STN_LDS_PTR(st_ + (2 * wave + j) * 1024), 16, offA[j], ko_, 0, 0); \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, STN_LDS_PTR(st_ + BM * RK * 2 + (2 * wave + j) * 1024), 16, offW[j], ko_, 0, 0); \
        }                                                                                                          \
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

    // prologue: up to three stages in flight
    if (0 < nk) STN_ISSUE(0);
    if (1 < nk) STN_ISSUE(1);
    if (2 < nk) STN_ISSUE(2);

    const int lr = lane & 31, lh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed once at most (stages still ahead) x 4 of this wave's DMAs remain outstanding
        const int ahead = nk - 1 - kt;
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave's pieces of stage kt are in LDS; stage kt-1 is fully consumed
        if (kt + 3 < nk) STN_ISSUE(kt + 3);  // refill the buffer that stage kt-1 just vacated
        const unsigned char* sa = smem + (kt & (RSTAGES - 1)) * RSTAGE_BYTES;
        const unsigned char* sb = sa + BM * RK * 2;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chunk = ks * 2 + lh;
            bf16x8 a[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int row = wm * 64 + mi * 32 + lr;
                a[mi] = *reinterpret_cast<const bf16x8*>(sa + row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int row = wn * 64 + ni * 32 + lr;
                b[ni] = *reinterpret_cast<const bf16x8*>(sb + row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4));
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
    }
#undef STN_ISSUE

    if (!VEC || MODE >= EPI_EULER_T) {
        run_epilogue<MODE>(e, acc, m0 + wm * 64, n0 + wn * 64, M, N, lane);
        return;
    }
    // ---- LDS-staged epilogue -------------------------------------------------------------------------
    __builtin_amdgcn_s_barrier();  // all fragment reads of the ring are done
    float* S = reinterpret_cast<float*>(smem);  // [128][128] fp32
    {
        const int half = lane >> 5, cl = lane & 31;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = wm * 64 + mi * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
                    S[row * BN + wn * 64 + ni * 32 + cl] = acc[mi][ni][i];
                }
    }
    __syncthreads();
    const int c8 = tid & 15;            // 8-column group inside the tile
    const int n = n0 + c8 * 8;
    if (n >= N) return;                 // whole column group out of range (N % 8 == 0 is guaranteed for VEC)
    float bias[8], gam[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { bias[j] = e.bias ? e.bias[n + j] : 0.f; gam[j] = (MODE == EPI_RESID && e.gamma) ? e.gamma[n + j] : 1.f; }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int row = (tid >> 4) + 16 * q;
        const int m = m0 + row;
        if (m >= M) continue;
        float keep = 1.f;
        if (e.len) { const int b = m / e.L; if (m - b * e.L >= e.len[b]) keep = 0.f; }
        const float4 v0 = *reinterpret_cast<const float4*>(S + row * BN + c8 * 8);
        const float4 v1 = *reinterpret_cast<const float4*>(S + row * BN + c8 * 8 + 4);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        const size_t o = (size_t)m * e.ldo + n;
        if (MODE == EPI_STORE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = act_f(v[j] + bias[j], e.act) * keep;
            if (e.out_dtype == BF16) {
                uint4 pk;
                pk.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
                pk.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
                pk.z = (unsigned)f2bf(v[4]) | ((unsigned)f2bf(v[5]) << 16);
                pk.w = (unsigned)f2bf(v[6]) | ((unsigned)f2bf(v[7]) << 16);
                *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(e.out) + o) = pk;
            } else {
                float* op = reinterpret_cast<float*>(e.out) + o;
                *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
        } else {  // EPI_RESID
            float* rp = e.resid + o;
            const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
            const float r[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (r[j] + gam[j] * (v[j] + bias[j])) * keep;
            *reinterpret_cast<float4*>(rp) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(rp + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}

void launch_gemm(hipStream_t s, int dtype, const void* A, int lda, const void* W, int ldw, int M, int N, int K,
                 const Epilogue& e) {
    if (M <= 0 || N <= 0) return;
    const int kq = dtype == BF16 ? 8 : 4;
    if (K <= 0 || K % kq || lda % kq || ldw % kq || (reinterpret_cast<uintptr_t>(A) & 15) ||
        (reinterpret_cast<uintptr_t>(W) & 15)) {
        fprintf(stderr, "stn: launch_gemm: operand shape/alignment violates the kernel contract (K=%d lda=%d ldw=%d)\n", K,
                lda, ldw);
        abort();
    }
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, ntiles = tiles_m * tiles_n;
    // v2 ring kernel: bf16, K % 32 == 0.  Vectorised epilogue needs 16-B aligned 8-column groups.
    const bool ring = dtype == BF16 && K % RK == 0;
    const void* optr = e.mode == EPI_RESID ? static_cast<const void*>(e.resid) : e.out;
    const bool vec = ring && e.mode <= EPI_RESID && N % 8 == 0 && e.ldo % 8 == 0 && !(reinterpret_cast<uintptr_t>(optr) & 15) &&
                     (!e.bias || !(reinterpret_cast<uintptr_t>(e.bias) & 3));
#define STN_LAUNCH(MODE)                                                                                         \
    if (ring && vec)                                                                                             \
        hipLaunchKernelGGL((gemm_bf16_ring_kernel<MODE, true>), dim3(ntiles), dim3(NT), 0, s,                    \
                           static_cast<const uint16_t*>(A), lda, static_cast<const uint16_t*>(W), ldw, M, N, K, tiles_n, ntiles, e); \
    else if (ring)                                                                                               \
        hipLaunchKernelGGL((gemm_bf16_ring_kernel<MODE, false>), dim3(ntiles), dim3(NT), 0, s,                   \
                           static_cast<const uint16_t*>(A), lda, static_cast<const uint16_t*>(W), ldw, M, N, K, tiles_n, ntiles, e); \
    else if (dtype == BF16)                                                                                      \
        hipLaunchKernelGGL(gemm_bf16_kernel<MODE>, dim3(ntiles), dim3(NT), 0, s, static_cast<const uint16_t*>(A), \
                           lda, static_cast<const uint16_t*>(W), ldw, M, N, K, tiles_n, ntiles, e);              \
    else                                                                                                         \
        hipLaunchKernelGGL(gemm_f32_kernel<MODE>, dim3(ntiles), dim3(NT), 0, s, static_cast<const float*>(A),     \
                           lda, static_cast<const float*>(W), ldw, M, N, K, tiles_n, ntiles, e);
    switch (e.mode) {
        case EPI_STORE: STN_LAUNCH(EPI_STORE) break;
        case EPI_RESID: STN_LAUNCH(EPI_RESID) break;
        case EPI_EULER_T: STN_LAUNCH(EPI_EULER_T) break;
        default: STN_LAUNCH(EPI_STORE_T) break;
    }
#undef STN_LAUNCH
}

}  // namespace stn
