// kernels_gemm.hip — MFMA GEMM with fused epilogues for gfx950 (MI355X).
//
//   acc[m][n] = sum_k A[m][k] * W[n][k]        A: activations [M][K], W: weights [N][K] (both K-contiguous)
//
// Used for every dense linear / pointwise conv of the four graphs that replace the reference's
// Ort::Session::Run sites (/root/reference/cpp/helper.cpp:519,552,643,668).
//
// Three generations live here; launch_gemm picks by shape:
//   gemm_tiled_kernel<MODE,BM,BN,WM,WN,NSTAGE,KS,ESZ>  (the product path; K % 32 == 0, vector epilogues)
//       operands HBM/L2 -> LDS by buffer_load ... lds (1 KiB per wave-instruction) into an NSTAGE-deep ring, counted
//       s_waitcnt vmcnt + raw s_barrier per K-step, ds_read_b128 fragments from an XOR-swizzled image,
//       v_mfma_f32_32x32x16_bf16 (bf16) or the exact v_mfma_f32_32x32x2_f32 (fp32).  Tile configurations and what each is for:
//       launch_tiled_auto.  Epilogues: (a) bf16 store through a wave-private transposed LDS image (ds_read_b64_tr_b16),
//       (b) fp32 slab transpose through the dead ring with 16-byte stores — bias, GELU/SiLU, layer-scale + residual,
//       row mask or packed-row sequence lookup, per-sequence time vector.
//   gemm_bf16_ring_kernel   128x128, K % 32 == 0 but epilogue operands not 16-byte aligned (rare).
//   gemm_bf16_kernel / gemm_f32_kernel   128x128, register-staged double buffer: any K % 8 (bf16) / % 4 (fp32), and the
//       transposing epilogues (EPI_STORE_T, EPI_EULER_T) of the host-pointer stages.
// Workgroup -> tile map is XCD-aware: the tiles that share an A row-panel are consecutive and land on
// one XCD (private 4 MiB L2), the bijective remap of the CDNA4 guide.
#include "kernels.hpp"
#include "dev_env.hpp"
#include "kernels_dev.hpp"

#include <hip/hip_bf16.h>
#include <stdio.h>
#include <stdlib.h>

namespace stn {

static constexpr int BM = 128, BN = 128, NT = 256;
static constexpr int STAGE_BYTES = (BM + BN) * 128;  // 32 KiB

// XCD-aware bijective remap of a 1-D grid (consecutive logical tiles -> same XCD).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + loc;
}

// One lane's share of the epilogue: accumulator tile (mi, ni) element i lives at
//   row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5), col = lane & 31      (32x32 MFMA C/D map)
template <int MODE, bool F16 = false>
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
    const bool need_bt = (e.len != nullptr) || MODE >= EPI_EULER_T || (MODE == EPI_RESID && e.rowvec != nullptr);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = m_base + mi * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
            if (m >= M) continue;
            int b = 0, t = m;
            float keep = 1.f;
            if (need_bt) {
                if (e.row_b) { b = e.row_b[m]; t = 0; }
                else {
                    b = m / e.L;
                    t = m - b * e.L;
                    if (e.len && t >= e.len[b]) keep = 0.f;
                }
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int n = ncol[ni];
                if (n >= N) continue;
                const float v = acc[mi][ni][i] + bias[ni];
                if (MODE == EPI_STORE) {
                    const float r = act_out_f(v, e.act, e.out_dtype == BF16) * keep;
                    const size_t o = (size_t)m * e.ldo + n;
                    if (e.out_dtype != F32) reinterpret_cast<uint16_t*>(e.out)[o] = cvt16<F16>(r);
                    else reinterpret_cast<float*>(e.out)[o] = r;
                } else if (MODE == EPI_RESID) {
                    const size_t o = (size_t)m * e.ldo + n;
                    const float rv = e.rowvec ? e.rowvec[(size_t)b * e.rv_ld + n] : 0.f;
                    e.resid[o] = (e.resid[o] + gam[ni] * v + rv) * keep;
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
template <int MODE, bool F16 = false>
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
                    acc[mi][ni] = mfma16<F16>(a[mi], b[ni], acc[mi][ni]);
        }
        if (more) STN_SWRITE((kt + 1) & 1);
        __syncthreads();
    }
#undef STN_GLOAD
#undef STN_SWRITE
    run_epilogue<MODE, F16>(e, acc, m0 + wm * 64, n0 + wn * 64, M, N, lane);
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


template <int MODE, bool VEC, bool F16 = false>
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
                    acc[mi][ni] = mfma16<F16>(a[mi], b[ni], acc[mi][ni]);
        }
    }
#undef STN_ISSUE

    if (!VEC || MODE >= EPI_EULER_T) {
        run_epilogue<MODE, F16>(e, acc, m0 + wm * 64, n0 + wn * 64, M, N, lane);
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
        const int b = e.row_b ? e.row_b[m] : ((e.len || (MODE == EPI_RESID && e.rowvec)) ? m / e.L : 0);
        if (!e.row_b && e.len && m - b * e.L >= e.len[b]) keep = 0.f;
        const float4 v0 = *reinterpret_cast<const float4*>(S + row * BN + c8 * 8);
        const float4 v1 = *reinterpret_cast<const float4*>(S + row * BN + c8 * 8 + 4);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        const size_t o = (size_t)m * e.ldo + n;
        if (MODE == EPI_STORE) {
            act8(v, bias, e.act, e.out_dtype == BF16, keep);
            if (e.out_dtype != F32) {
                uint4 pk;
                pk.x = (unsigned)cvt16<F16>(v[0]) | ((unsigned)cvt16<F16>(v[1]) << 16);
                pk.y = (unsigned)cvt16<F16>(v[2]) | ((unsigned)cvt16<F16>(v[3]) << 16);
                pk.z = (unsigned)cvt16<F16>(v[4]) | ((unsigned)cvt16<F16>(v[5]) << 16);
                pk.w = (unsigned)cvt16<F16>(v[6]) | ((unsigned)cvt16<F16>(v[7]) << 16);
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
            float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0;
            if (e.rowvec) { const float* tp = e.rowvec + (size_t)b * e.rv_ld + n; t0 = *reinterpret_cast<const float4*>(tp); t1 = *reinterpret_cast<const float4*>(tp + 4); }
            const float tv[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (r[j] + gam[j] * (v[j] + bias[j]) + tv[j]) * keep;
            *reinterpret_cast<float4*>(rp) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(rp + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}


// ---------------------------------------------------------------------------------------------
// bf16 v3: the ring kernel generalised over the tile shape.  Per-CU operand ingest (L2/HBM -> LDS), not MFMA
// issue, bounds the 128x128 tile (64 FLOP per ingested byte): a BM x BN tile needs (BM+BN)*64 B per 32-deep
// K-step for 2*BM*BN*32 FLOP, so 256x256 halves the bytes per FLOP and 256x128 cuts them by a quarter.
//   WM x WN wavefronts, each owning (TM*32) x (TN*32) of the tile; NSTAGE-deep LDS ring, NSTAGE-1 stages in flight.
//   Epilogue: TM passes; pass mi stages the mi-th 32-row slab of every wave row ([WM*32][BN] fp32) through LDS and
//   writes it out with 16-B vectors (STORE / RESID only, N % 8 == 0).
// ---------------------------------------------------------------------------------------------
template <int MODE, int BM_, int BN_, int WM, int WN, int NSTAGE, int KS, int ESZ, bool F16 = false>
__global__ __launch_bounds__(WM* WN * 64) void gemm_tiled_kernel(const void* __restrict__ Av, int lda,
                                                                   const void* __restrict__ Wv, int ldw, int M, int N, int K,
                                                                   int tiles_n, int ntiles, Epilogue e) {
    // ESZ = operand element size: 2 (bf16, v_mfma_f32_32x32x16_bf16) or 4 (fp32, exact v_mfma_f32_32x32x2_f32)
    // KS  = K elements per stage; rows of KS*ESZ = 64 or 128 bytes (128 = full cache lines per DMA row segment)
    constexpr int NW = WM * WN, NTHR = NW * 64;
    constexpr int TM = BM_ / WM / 32, TN = BN_ / WN / 32;
    constexpr int ROWB = KS * ESZ;               // bytes per LDS row
    constexpr int RPP = 1024 / ROWB;             // rows per 1-KiB DMA piece (16 or 8)
    constexpr int CPR = ROWB / 16;               // 16-B chunks per row (4 or 8)
    constexpr int EPC = 16 / ESZ;                // elements per 16-B chunk
    constexpr int PA = BM_ / RPP / NW, PW = BN_ / RPP / NW, PER = PA + PW;  // DMA pieces per wave per stage
    constexpr int STAGE = (BM_ + BN_) * ROWB;
    static_assert(ROWB == 64 || ROWB == 128, "row bytes");
    // GEN: the 1-KiB DMA pieces of a stage do not divide evenly among the waves (3/4-size tiles: 192 or 96 rows on 12 or 6
    // waves).  Then wave w takes pieces w, w + NW, ... of the whole stage (A rows first, W rows after), and the slots past
    // the last piece become zero-byte loads into a per-wave scratch KiB so that every wave's vmcnt arithmetic stays uniform.
    constexpr bool GEN = (BM_ / RPP) % NW != 0 || (BN_ / RPP) % NW != 0;
    constexpr int PT = (BM_ + BN_) / RPP;                 // pieces per stage
    constexpr int PERG = (PT + NW - 1) / NW;              // slots per wave and stage in the GEN scheme
    static_assert(BM_ % (WM * 32) == 0 && BN_ % (WN * 32) == 0 && BM_ % RPP == 0 && BN_ % RPP == 0, "tile/wave shape");
    static_assert(NSTAGE * STAGE >= WM * 32 * BN_ * 4, "epilogue slab must fit in the ring");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // NSTAGE * STAGE bytes, the only LDS object
    const unsigned char* A = static_cast<const unsigned char*>(Av);
    const unsigned char* W = static_cast<const unsigned char*>(Wv);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // split-K (e.ksplit > 1): workgroup blockIdx.x = split * ntiles + tile sums its own K range into the fp32 workspace
    // e.out + split * M * N (plain store); splitk_reduce_kernel adds the splits in a fixed order and applies the real epilogue
    const int ksp = e.ksplit > 1 ? e.ksplit : 1;
    const int split = ksp > 1 ? (int)blockIdx.x / ntiles : 0;
    const int tile = xcd_remap((int)blockIdx.x - split * ntiles, ntiles);
    if (ksp > 1) e.out = static_cast<float*>(e.out) + (size_t)split * M * N;
    const int m0 = (tile / tiles_n) * BM_, n0 = (tile % tiles_n) * BN_;
    unsigned long long t_in = 0, t_first = 0, t_loop = 0;
    if (e.ts) t_in = __builtin_readcyclecounter();

    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A + (size_t)m0 * lda * ESZ, m0 < M ? (size_t)(M - m0) * lda * ESZ : 0);
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(W + (size_t)n0 * ldw * ESZ, n0 < N ? (size_t)(N - n0) * ldw * ESZ : 0);
    // swizzle of the 16-B slot inside a row: 64-B rows use (row>>2)&3, 128-B rows (row>>1)&7 (see the bank analysis above)
    auto swz = [](int row) { return ROWB == 64 ? ((row >> 2) & 3) : ((row >> 1) & 7); };
    constexpr int PER_ = GEN ? PERG : PER;
    unsigned offA[GEN ? 1 : PA], offW[GEN ? 1 : PW];
    unsigned offG[GEN ? PERG : 1];
    int kindG[GEN ? PERG : 1];  // 0 = A piece, 1 = W piece, 2 = padding slot (wave-uniform)
    __amdgpu_buffer_rsrc_t rsZ = rsA;
    if constexpr (GEN) {
        rsZ = make_rsrc(A, 0);  // zero-length range: every lane out of range, no memory traffic
#pragma unroll
        for (int j = 0; j < PERG; ++j) {
            const int p = wave + NW * j;  // wave-uniform
            const int row = p * RPP + lane / CPR;
            if (p < BM_ / RPP) {
                kindG[j] = 0;
                offG[j] = (unsigned)(row * lda + (((lane % CPR) ^ swz(row)) * EPC)) * (unsigned)ESZ;
            } else if (p < PT) {
                const int rw = row - BM_;
                kindG[j] = 1;
                offG[j] = (unsigned)(rw * ldw + (((lane % CPR) ^ swz(rw)) * EPC)) * (unsigned)ESZ;
            } else {
                kindG[j] = 2;
                offG[j] = 0u;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const int row = (PA * wave + j) * RPP + lane / CPR;
            offA[j] = (unsigned)(row * lda + (((lane % CPR) ^ swz(row)) * EPC)) * (unsigned)ESZ;
        }
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int row = (PW * wave + j) * RPP + lane / CPR;
            offW[j] = (unsigned)(row * ldw + (((lane % CPR) ^ swz(row)) * EPC)) * (unsigned)ESZ;
        }
    }
    const int nk_all = K / KS;
    const int kt0 = split * (nk_all / ksp), nk = ksp > 1 ? kt0 + nk_all / ksp : nk_all;  // this workgroup's K-steps [kt0, nk)

#define STN_ISSUE(kt)                                                                                               \
    {                                                                                                               \
        unsigned char* st_ = smem + ((kt) % NSTAGE) * STAGE;                                                        \
        const int ko_ = (kt) * ROWB;                                                                                \
        if constexpr (GEN) {                                                                                        \
            _Pragma("unroll") for (int j = 0; j < PERG; ++j) {                                                      \
                if (kindG[j] == 0) dma16(rsA, st_ + (wave + NW * j) * 1024, offG[j], ko_);                          \
                else if (kindG[j] == 1) dma16(rsW, st_ + (wave + NW * j) * 1024, offG[j], ko_);                     \
                else dma16(rsZ, smem + NSTAGE * STAGE + wave * 1024, 0u, 0);                                        \
            }                                                                                                       \
        } else {                                                                                                    \
            _Pragma("unroll") for (int j = 0; j < PA; ++j)                                                          \
                dma16(rsA, st_ + (PA * wave + j) * 1024, offA[j], ko_);                                              \
            _Pragma("unroll") for (int j = 0; j < PW; ++j)                                                          \
                dma16(rsW, st_ + BM_ * ROWB + (PW * wave + j) * 1024, offW[j], ko_);                                 \
        }                                                                                                           \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

#pragma unroll
    for (int st = 0; st < NSTAGE - 1; ++st)
        if (kt0 + st < nk) STN_ISSUE(kt0 + st);

    const int lr = lane & 31, lh = lane >> 5;
    for (int kt = kt0; kt < nk; ++kt) {
        wait_stage<PER_, NSTAGE - 2>(nk - 1 - kt);
        __builtin_amdgcn_s_barrier();
        if (e.ts && kt == kt0) t_first = __builtin_readcyclecounter();
        const unsigned char* sa = smem + (kt % NSTAGE) * STAGE;
        const unsigned char* sb = sa + BM_ * ROWB;
        if constexpr (ESZ == 2) {
            // The K-loop is latency-bound per wave, not bandwidth-bound (PMC: L1->L2 read latency ~420 cycles, MFMA pipe busy
            // ~30 %): so ALL fragment reads of the stage are issued right behind the barrier, the LDS-DMA refill of the
            // vacated stage goes out while they are in flight, and the MFMAs then run back to back behind counted lgkmcnt waits.
            constexpr int NKS = KS / 16;
            bf16x8 a[NKS][TM], b[NKS][TN];
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int chunk = ks * 2 + lh;
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
                    const int row = (wm * TM + mi) * 32 + lr;
                    a[ks][mi] = *reinterpret_cast<const bf16x8*>(sa + row * ROWB + ((chunk ^ swz(row)) << 4));
                }
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    const int row = (wn * TN + ni) * 32 + lr;
                    b[ks][ni] = *reinterpret_cast<const bf16x8*>(sb + row * ROWB + ((chunk ^ swz(row)) << 4));
                }
            }
            if (kt + NSTAGE - 1 < nk) STN_ISSUE(kt + NSTAGE - 1);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[mi][ni] = mfma16<F16>(a[ks][mi], b[ks][ni], acc[mi][ni]);
        } else {
            if (kt + NSTAGE - 1 < nk) STN_ISSUE(kt + NSTAGE - 1);
            // fp32: lane (row r, half h) feeds A[r][k = 2*ks + h]; one ds_read_b128 covers the lane's k for two MFMA steps
#pragma unroll
            for (int kq = 0; kq < KS / 4; ++kq) {  // 4 consecutive k per 16-B chunk: steps 2kq (k = 4kq + h) and 2kq+1 (k = 4kq + 2 + h)
                float4 a4[TM], b4[TN];
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
                    const int row = (wm * TM + mi) * 32 + lr;
                    a4[mi] = *reinterpret_cast<const float4*>(sa + row * ROWB + ((kq ^ swz(row)) << 4));
                }
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    const int row = (wn * TN + ni) * 32 + lr;
                    b4[ni] = *reinterpret_cast<const float4*>(sb + row * ROWB + ((kq ^ swz(row)) << 4));
                }
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(lh ? a4[mi].y : a4[mi].x, lh ? b4[ni].y : b4[ni].x, acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(lh ? a4[mi].w : a4[mi].z, lh ? b4[ni].w : b4[ni].z, acc[mi][ni], 0, 0, 0);
                    }
            }
        }
    }
#undef STN_ISSUE
    if (e.ts) t_loop = __builtin_readcyclecounter();

    // ---- bf16 store epilogue through a wave-private transposed image (gfx950 ds_read_b64_tr_b16) --------------------------
    // In the accumulator layout a lane holds 4 CONSECUTIVE ROWS of one column per register quad.  After bias + activation
    // those 4 values are packed to 8 bytes and written with one ds_write_b64 into a [column][row] image of the wave's own
    // 32x32 subtile (72-byte rows: conflict-free); the hardware transpose read hands each lane 4 consecutive COLUMNS of one
    // row, two of them make the 16-byte global store.  Against the fp32 slab transpose below: 4 LDS writes per subtile and
    // lane instead of 16, half the read bytes, and no workgroup barrier — every wave drains its tiles at its own pace.
    if (MODE == EPI_STORE && ESZ == 2 && e.out_dtype != F32 && e.len == nullptr && e.tr_epilogue) {
        constexpr int TSTR = 72, TIMG = 32 * TSTR;  // bytes per image row / per image
        static_assert(NSTAGE * STAGE >= NW * 2 * TIMG, "transposed images must fit in the ring");
        typedef short v4s_ __attribute__((ext_vector_type(4)));
        __syncthreads();  // every wave is done reading the operand ring
        unsigned char* img0 = smem + wave * (2 * TIMG);
        const int cl_ = lane & 31, hf_ = lane >> 5;
        const int G_ = lane >> 4, i16_ = lane & 15, q_ = i16_ >> 2, p_ = i16_ & 3;
        int buf = 0;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int ncol = n0 + (wn * TN + ni) * 32 + cl_;
            const float bs = (ncol < N && e.bias) ? e.bias[ncol] : 0.f;
            const int nst = n0 + (wn * TN + ni) * 32 + 8 * G_;  // first of this lane's 8 output columns
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                unsigned char* img = img0 + buf * TIMG;
                buf ^= 1;
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = acc[mi][ni][i] + bs;
                if (e.act == ACT_GELU) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = F16 ? gelu_f(v[i]) : gelu_bf16_f(v[i]);  // half keeps 11 bits: erf form
                } else if (e.act == ACT_GELU_TANH) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = F16 ? gelu_tanh_f(v[i]) : gelu_bf16_f(v[i]);
                } else if (e.act == ACT_SILU) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = v[i] / (1.0f + expf(-v[i]));
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 pk;
                    pk.x = (unsigned)cvt16<F16>(v[4 * g]) | ((unsigned)cvt16<F16>(v[4 * g + 1]) << 16);
                    pk.y = (unsigned)cvt16<F16>(v[4 * g + 2]) | ((unsigned)cvt16<F16>(v[4 * g + 3]) << 16);
                    *reinterpret_cast<uint2*>(img + cl_ * TSTR + (8 * g + 4 * hf_) * 2) = pk;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // LDS is in order per wave; keep the compiler in order too
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    // block rows = tile columns 8G+q (and +4), block columns = tile rows 16t + 4p ..; lane i16 receives tile row 16t+i16
                    const unsigned char* a0 = img + (8 * G_ + q_) * TSTR + (16 * t + 4 * p_) * 2;
                    const v4s_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_*)(a0));
                    const v4s_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_*)(a0 + 4 * TSTR));
                    const int m = m0 + (wm * TM + mi) * 32 + 16 * t + i16_;
                    if (m < M && nst < N) {
                        uint4 o;
                        o.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
                        o.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
                        o.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
                        o.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
                        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(e.out) + (size_t)m * e.ldo + nst) = o;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
        }
        if (e.ts && tid == 0) {
            __builtin_amdgcn_s_waitcnt(0);
            unsigned long long* tp = e.ts + (size_t)blockIdx.x * 4;
            tp[0] = t_in; tp[1] = t_first; tp[2] = t_loop; tp[3] = __builtin_readcyclecounter();
        }
        return;
    }

    // ---- epilogue: TM passes over [WM*32][BN] fp32 slabs --------------------------------------------------
    // A thread keeps ONE 8-column group for the whole epilogue (NTHR is a multiple of BN/8), so bias / layer-scale
    // are loaded once, and the per-pass loop is fully unrolled with its global loads (residual) issued up front.
    float* S = reinterpret_cast<float*>(smem);
    constexpr int CG = BN_ / 8;                 // 8-column groups per row
    constexpr int ROWS_PER_IT = NTHR / CG;      // slab rows covered by one sweep of the workgroup
    constexpr int ITER = (WM * 32) / ROWS_PER_IT;
    static_assert(NTHR % CG == 0 && (WM * 32) % ROWS_PER_IT == 0, "epilogue thread map");
    const int half = lane >> 5, cl = lane & 31;
    const int c8 = tid % CG, rbase = tid / CG;
    const int n = n0 + c8 * 8;
    const bool ncol_ok = n < N;
    float bias[8], gam[8];
    {
        float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0, g0 = make_float4(1.f, 1.f, 1.f, 1.f), g1 = g0;
        if (ncol_ok && e.bias) { b0 = *reinterpret_cast<const float4*>(e.bias + n); b1 = *reinterpret_cast<const float4*>(e.bias + n + 4); }
        if (MODE == EPI_RESID && ncol_ok && e.gamma) { g0 = *reinterpret_cast<const float4*>(e.gamma + n); g1 = *reinterpret_cast<const float4*>(e.gamma + n + 4); }
        bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w; bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
        gam[0] = g0.x; gam[1] = g0.y; gam[2] = g0.z; gam[3] = g0.w; gam[4] = g1.x; gam[5] = g1.y; gam[6] = g1.z; gam[7] = g1.w;
    }
    const bool to_bf16 = e.out_dtype == BF16;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        // residual rows of this pass: issue the loads before the LDS hand-off so their latency hides behind it
        float4 r0[ITER], r1[ITER];
        bool ok[ITER];
        float keep[ITER];
        size_t off[ITER];
        int bq[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int srow = rbase + it * ROWS_PER_IT;
            const int m = m0 + ((srow >> 5) * TM + mi) * 32 + (srow & 31);
            ok[it] = ncol_ok && m < M;
            keep[it] = 1.f;
            off[it] = (size_t)m * e.ldo + n;
            bq[it] = 0;
            if (ok[it] && e.row_b) {
                bq[it] = e.row_b[m];  // packed rows: every row is valid, only the sequence index is needed
            } else if (ok[it] && (e.len || (MODE == EPI_RESID && e.rowvec))) {
                const int b = m / e.L;
                bq[it] = b;
                if (e.len && m - b * e.L >= e.len[b]) keep[it] = 0.f;
            }
            if (MODE == EPI_RESID) {
                r0[it] = make_float4(0.f, 0.f, 0.f, 0.f); r1[it] = r0[it];
                if (ok[it]) { r0[it] = *reinterpret_cast<const float4*>(e.resid + off[it]); r1[it] = *reinterpret_cast<const float4*>(e.resid + off[it] + 4); }
            }
        }
        __syncthreads();  // ring (pass 0) or previous slab (later passes) fully consumed
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int srow = wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
                S[srow * BN_ + (wn * TN + ni) * 32 + cl] = acc[mi][ni][i];
            }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int srow = rbase + it * ROWS_PER_IT;
            const float4 v0 = *reinterpret_cast<const float4*>(S + srow * BN_ + c8 * 8);
            const float4 v1 = *reinterpret_cast<const float4*>(S + srow * BN_ + c8 * 8 + 4);
            if (!ok[it]) continue;
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            if (MODE == EPI_STORE) {
                act8(v, bias, e.act, to_bf16, keep[it]);
                if (e.out_dtype != F32) {
                    uint4 pk;
                    pk.x = (unsigned)cvt16<F16>(v[0]) | ((unsigned)cvt16<F16>(v[1]) << 16);
                    pk.y = (unsigned)cvt16<F16>(v[2]) | ((unsigned)cvt16<F16>(v[3]) << 16);
                    pk.z = (unsigned)cvt16<F16>(v[4]) | ((unsigned)cvt16<F16>(v[5]) << 16);
                    pk.w = (unsigned)cvt16<F16>(v[6]) | ((unsigned)cvt16<F16>(v[7]) << 16);
                    uint4* po = reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(e.out) + off[it]);
                    if (e.nt & 1) { const u32x4 pv = {pk.x, pk.y, pk.z, pk.w}; __builtin_nontemporal_store(pv, reinterpret_cast<u32x4*>(po)); }
                    else *po = pk;
                } else {
                    float* op = reinterpret_cast<float*>(e.out) + off[it];
                    *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
            } else {  // EPI_RESID
                const float r[8] = {r0[it].x, r0[it].y, r0[it].z, r0[it].w, r1[it].x, r1[it].y, r1[it].z, r1[it].w};
                if (e.rowvec) {  // wave-uniform branch; the vector is a few KB shared by ~L rows: L1/L2 hits
                    const float* tp = e.rowvec + (size_t)bq[it] * e.rv_ld + n;
                    const float4 t0 = *reinterpret_cast<const float4*>(tp), t1 = *reinterpret_cast<const float4*>(tp + 4);
                    const float tv[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (r[j] + gam[j] * (v[j] + bias[j]) + tv[j]) * keep[it];
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (r[j] + gam[j] * (v[j] + bias[j])) * keep[it];
                }
                float* rp = e.resid + off[it];
                *reinterpret_cast<float4*>(rp) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(rp + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
    }
    if (e.ts && tid == 0) {
        // the stores above are fire-and-forget: wait for them so the last stamp covers the epilogue's memory time too
        __builtin_amdgcn_s_waitcnt(0);
        unsigned long long* tp = e.ts + (size_t)blockIdx.x * 4;
        tp[0] = t_in; tp[1] = t_first; tp[2] = t_loop; tp[3] = __builtin_readcyclecounter();
    }
}

template <int MODE, int BM_, int BN_, int WM, int WN, int NSTAGE, int KS, int ESZ = 2, bool F16 = false>
static void launch_tiled(hipStream_t s, const void* A, int lda, const void* W, int ldw, int M, int N, int K, const Epilogue& e) {
    constexpr int RPP_ = 1024 / (KS * ESZ), NW_ = WM * WN;
    constexpr bool GEN_ = (BM_ / RPP_) % NW_ != 0 || (BN_ / RPP_) % NW_ != 0;
    constexpr size_t lds = (size_t)NSTAGE * (BM_ + BN_) * KS * ESZ + (GEN_ ? (size_t)NW_ * 1024 : 0);  // + padding-slot scratch
    static PerDeviceOnce attr_once;
    if (attr_once.need()) {
        stn_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tiled_kernel<MODE, BM_, BN_, WM, WN, NSTAGE, KS, ESZ, F16>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "hipFuncSetAttribute(gemm_tiled)");
    }
    const int tiles_m = (M + BM_ - 1) / BM_, tiles_n = (N + BN_ - 1) / BN_, ntiles = tiles_m * tiles_n;
    const int ksp = e.ksplit > 1 ? e.ksplit : 1;
    if (ksp > 1 && (MODE != EPI_STORE || (K / KS) % ksp != 0)) { throw std::invalid_argument("split-K needs a plain store epilogue and K/KS divisible by the split"); }
    STN_KLAUNCH((gemm_tiled_kernel<MODE, BM_, BN_, WM, WN, NSTAGE, KS, ESZ, F16>), dim3(ntiles * ksp), dim3(WM * WN * 64), lds, s, A, lda,
                       W, ldw, M, N, K, tiles_n, ntiles, e);
}

// tile-shape selection for the vectorised-epilogue path; STN_GEMM_CFG=<n> forces one bf16 shape (experiments)
static int g_gemm_cfg = -2;
template <int MODE>
static bool launch_tiled_auto(hipStream_t s, int dtype, const void* A, int lda, const void* W, int ldw, int M, int N, int K,
                              const Epilogue& e) {
    if (dtype == F32) {
        // fp32 MFMA is 1/16 of the bf16 rate: always compute-bound, tile choice only has to keep the CUs busy
        if (K % 32) return false;
        // (a narrow N leaves few 128-wide tiles: the duration predictor's 9 k x 128 pw2 is 71 of them on 256 CUs)
        if (M <= 64 || (long)((M + 127) / 128) * ((N + 127) / 128) < 160) launch_tiled<MODE, 64, 64, 2, 2, 4, 32, 4>(s, A, lda, W, ldw, M, N, K, e);
        else launch_tiled<MODE, 128, 128, 2, 2, 3, 32, 4>(s, A, lda, W, ldw, M, N, K, e);
        return true;
    }
    if (g_gemm_cfg == -2) { const char* c = stn::dev_env("STN_GEMM_CFG"); g_gemm_cfg = c ? atoi(c) : -1; }
    int cfg = g_gemm_cfg;
    if (cfg < 0) {
        // measured on MI355X (tools/gemm_bench.py): the 256x256 tile halves the operand bytes per FLOP and wins whenever
        // it still yields ~a full wave of workgroups (1 per CU); otherwise the 128x128 tile with 128-byte rows keeps more
        // CUs busy; tiny M (single utterances) gets 64x64 tiles so that N is spread over more CUs.
        const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
        if (N >= 256 && t256 >= 160) {  // (a packed batch leaves ~180 of these tiles: still better than 700 small ones)
            const long t192 = (long)((M + 191) / 192) * ((N + 255) / 256);
            if (t256 < 208 && t192 <= 256) cfg = 18;       // one thin round (packed batches): 192-row tiles refill the idle CUs (-7 %)
            else if (t256 < 512) cfg = 11;                 // one round of tiles: 16 waves shorten the per-tile critical path
            else if (K <= 512 && t256 >= 1024) cfg = 17;   // short K, many tiles: 256x128, 8 waves, 2 WGs/CU (4 % over the 4-wave form)
            else cfg = 1;
        } else if (K % 64 == 0) cfg = M <= 64 ? 12 : 8;
        else return false;
    }
    if ((cfg == 5 || cfg == 6 || cfg == 7 || cfg == 8 || cfg == 12 || cfg == 13 || cfg == 14) && K % 64) return false;
    static int g_tr = -2;
    if (g_tr == -2) { const char* c = stn::dev_env("STN_GEMM_TR"); g_tr = c ? atoi(c) : 1; }
    Epilogue et = e;
    // the transposed-image epilogue stores 64-byte row segments (16 rows per instruction): a win where a CU runs one tile
    // (-4 % ve.pw1, -15 % te.pw1), a loss where a co-resident workgroup's K loop competes for the vector-memory path (vo.pw1)
    et.tr_epilogue = g_tr == 2 || (g_tr == 1 && (cfg == 11 || cfg == 8 || cfg == 12 || cfg == 13 || cfg == 14 || cfg == 18));
    const Epilogue& e_ = et;
    if (dtype == F16) {  // the shapes the heuristic above picks; STN_GEMM_CFG experiments stay bf16-only
        switch (cfg) {
            case 1: launch_tiled<MODE, 256, 256, 2, 4, 4, 32, 2, true>(s, A, lda, W, ldw, M, N, K, e_); return true;
            case 8: launch_tiled<MODE, 128, 128, 2, 4, 4, 64, 2, true>(s, A, lda, W, ldw, M, N, K, e_); return true;
            case 11: launch_tiled<MODE, 256, 256, 4, 4, 4, 32, 2, true>(s, A, lda, W, ldw, M, N, K, e_); return true;
            case 12: launch_tiled<MODE, 64, 64, 2, 2, 4, 64, 2, true>(s, A, lda, W, ldw, M, N, K, e_); return true;
            case 17: launch_tiled<MODE, 256, 128, 4, 2, 3, 32, 2, true>(s, A, lda, W, ldw, M, N, K, e_); return true;
            case 18: launch_tiled<MODE, 192, 256, 3, 4, 4, 32, 2, true>(s, A, lda, W, ldw, M, N, K, e_); return true;
            default: return false;
        }
    }
    switch (cfg) {
        case 1: launch_tiled<MODE, 256, 256, 2, 4, 4, 32>(s, A, lda, W, ldw, M, N, K, e_); return true;
        case 2: launch_tiled<MODE, 256, 128, 4, 2, 5, 32>(s, A, lda, W, ldw, M, N, K, e_); return true;
        case 3: launch_tiled<MODE, 128, 128, 2, 2, 8, 32>(s, A, lda, W, ldw, M, N, K, e_); return true;
        case 4: launch_tiled<MODE, 128, 256, 2, 4, 5, 32>(s, A, lda, W, ldw, M, N, K, e_); return true;
        case 5: launch_tiled<MODE, 128, 128, 2, 2, 4, 64>(s, A, lda, W, ldw, M, N, K, e_); return true;
        case 6: launch_tiled<MODE, 128, 128, 2, 2, 3, 64>(s, A, lda, W, ldw, M, N, K, e_); return true;
        case 7: launch_tiled<MODE, 256, 128, 4, 2, 3, 64>(s, A, lda, W, ldw, M, N, K, e_); return true;
        case 8: launch_tiled<MODE, 128, 128, 2, 4, 4, 64>(s, A, lda, W, ldw, M, N, K, e_); return true;
        case 9: launch_tiled<MODE, 128, 256, 2, 2, 3, 32>(s, A, lda, W, ldw, M, N, K, e_); return true;   // 72 KiB: 2 WGs / CU
        case 10: launch_tiled<MODE, 256, 128, 2, 2, 3, 32>(s, A, lda, W, ldw, M, N, K, e_); return true;  // 72 KiB: 2 WGs / CU
        case 11: launch_tiled<MODE, 256, 256, 4, 4, 4, 32>(s, A, lda, W, ldw, M, N, K, e_); return true;  // 16 waves
        case 12: launch_tiled<MODE, 64, 64, 2, 2, 4, 64>(s, A, lda, W, ldw, M, N, K, e_); return true;    // tiny M
        case 13: launch_tiled<MODE, 128, 128, 4, 4, 4, 64>(s, A, lda, W, ldw, M, N, K, e_); return true;  // 16 waves, 4 per SIMD
        case 14: launch_tiled<MODE, 128, 64, 4, 2, 4, 64>(s, A, lda, W, ldw, M, N, K, e_); return true;   // more tiles for narrow N
        case 15: launch_tiled<MODE, 128, 128, 2, 2, 3, 32>(s, A, lda, W, ldw, M, N, K, e_); return true;  // 48 KiB: 3 WGs / CU
        case 16: launch_tiled<MODE, 128, 256, 2, 4, 3, 32>(s, A, lda, W, ldw, M, N, K, e_); return true;  // 72 KiB, 8 waves: 2 WGs / CU
        case 17: launch_tiled<MODE, 256, 128, 4, 2, 3, 32>(s, A, lda, W, ldw, M, N, K, e_); return true;  // 72 KiB, 8 waves: 2 WGs / CU
        case 18: launch_tiled<MODE, 192, 256, 3, 4, 4, 32>(s, A, lda, W, ldw, M, N, K, e_); return true;  // 12 waves: 3/4 of config 11, same work per wave
        default: return false;
    }
}

void launch_gemm(hipStream_t s, int dtype, const void* A, int lda, const void* W, int ldw, int M, int N, int K,
                 const Epilogue& e) {
    if (M <= 0 || N <= 0) return;
    const int kq = is_half(dtype) ? 8 : 4;
    if (K <= 0 || K % kq || lda % kq || ldw % kq || (reinterpret_cast<uintptr_t>(A) & 15) ||
        (reinterpret_cast<uintptr_t>(W) & 15)) { char m_[256]; snprintf(m_, sizeof m_, "launch_gemm: operand shape/alignment violates the kernel contract (K=%d lda=%d ldw=%d)", K,
                lda, ldw); throw std::invalid_argument(m_); }
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, ntiles = tiles_m * tiles_n;
    // v2 ring kernel: bf16, K % 32 == 0.  Vectorised epilogue needs 16-B aligned 8-column groups.
    const bool ring = is_half(dtype) && K % RK == 0;
    const void* optr = e.mode == EPI_RESID ? static_cast<const void*>(e.resid) : e.out;
    const bool vec_ok = e.mode <= EPI_RESID && N % 8 == 0 && e.ldo % 8 == 0 && !(reinterpret_cast<uintptr_t>(optr) & 15) &&
                        (!e.bias || !(reinterpret_cast<uintptr_t>(e.bias) & 15)) && (!e.gamma || !(reinterpret_cast<uintptr_t>(e.gamma) & 15));
    const bool vec = ring && vec_ok;
    if (vec_ok && (ring || dtype == F32)) {
        if (e.mode == EPI_STORE && launch_tiled_auto<EPI_STORE>(s, dtype, A, lda, W, ldw, M, N, K, e)) return;
        if (e.mode == EPI_RESID && launch_tiled_auto<EPI_RESID>(s, dtype, A, lda, W, ldw, M, N, K, e)) return;
    }
#define STN_LAUNCH_H(MODE, F16_)                                                                                  \
    if (ring && vec)                                                                                             \
        STN_KLAUNCH((gemm_bf16_ring_kernel<MODE, true, F16_>), dim3(ntiles), dim3(NT), 0, s,                    \
                           static_cast<const uint16_t*>(A), lda, static_cast<const uint16_t*>(W), ldw, M, N, K, tiles_n, ntiles, e); \
    else if (ring)                                                                                               \
        STN_KLAUNCH((gemm_bf16_ring_kernel<MODE, false, F16_>), dim3(ntiles), dim3(NT), 0, s,                   \
                           static_cast<const uint16_t*>(A), lda, static_cast<const uint16_t*>(W), ldw, M, N, K, tiles_n, ntiles, e); \
    else                                                                                                         \
        STN_KLAUNCH((gemm_bf16_kernel<MODE, F16_>), dim3(ntiles), dim3(NT), 0, s, static_cast<const uint16_t*>(A), \
                           lda, static_cast<const uint16_t*>(W), ldw, M, N, K, tiles_n, ntiles, e);
#define STN_LAUNCH(MODE)                                                                                         \
    if (dtype == F16) { STN_LAUNCH_H(MODE, true) }                                                               \
    else if (dtype == BF16) { STN_LAUNCH_H(MODE, false) }                                                        \
    else                                                                                                         \
        STN_KLAUNCH(gemm_f32_kernel<MODE>, dim3(ntiles), dim3(NT), 0, s, static_cast<const float*>(A),     \
                           lda, static_cast<const float*>(W), ldw, M, N, K, tiles_n, ntiles, e);
    switch (e.mode) {
        case EPI_STORE: STN_LAUNCH(EPI_STORE) break;
        case EPI_RESID: STN_LAUNCH(EPI_RESID) break;
        case EPI_EULER_T: STN_LAUNCH(EPI_EULER_T) break;
        default: STN_LAUNCH(EPI_STORE_T) break;
    }
#undef STN_LAUNCH
#undef STN_LAUNCH_H
}

// ---------------------------------------------------------------------------------------------
// split-K reduction: out = epilogue(sum over splits, in split order, of the fp32 partials).  Tiny-M GEMMs with a long K
// (a single utterance's 49 x 384 x 1536 in exact fp32) otherwise run 48 K-steps on 6 workgroups.
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ void splitk_reduce_kernel(const float* __restrict__ part, int S, int M, int N, Epilogue e) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over [M][N/4]
    const int N4 = N >> 2;
    if (i >= (int64_t)M * N4) return;
    const int m = (int)(i / N4), n = (int)(i - (int64_t)m * N4) * 4;
    float4 a = reinterpret_cast<const float4*>(part)[i];
    for (int s2 = 1; s2 < S; ++s2) {
        const float4 b = reinterpret_cast<const float4*>(part + (size_t)s2 * M * N)[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float v[4] = {a.x, a.y, a.z, a.w};
    float keep = 1.f;
    int bsel = 0;
    if (e.row_b) bsel = e.row_b[m];
    else if (e.len || e.rowvec) { bsel = m / e.L; if (e.len && m - bsel * e.L >= e.len[bsel]) keep = 0.f; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float t = v[j] + (e.bias ? e.bias[n + j] : 0.f);
        if (MODE == EPI_STORE) {
            v[j] = act_out_f(t, e.act, e.out_dtype == BF16) * keep;
        } else {
            const float g = e.gamma ? e.gamma[n + j] : 1.f;
            const float rv = e.rowvec ? e.rowvec[(size_t)bsel * e.rv_ld + n + j] : 0.f;
            const size_t o = (size_t)m * e.ldo + n + j;
            v[j] = e.rowvec ? (e.resid[o] + g * t + rv) * keep : (e.resid[o] + g * t) * keep;
        }
    }
    const size_t o = (size_t)m * e.ldo + n;
    if (MODE == EPI_RESID) {
        *reinterpret_cast<float4*>(e.resid + o) = make_float4(v[0], v[1], v[2], v[3]);
    } else if (e.out_dtype == BF16) {
        uint2 pk;
        pk.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
        pk.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(e.out) + o) = pk;
    } else if (e.out_dtype == F16) {
        uint2 pk;
        pk.x = (unsigned)cvt16<true>(v[0]) | ((unsigned)cvt16<true>(v[1]) << 16);
        pk.y = (unsigned)cvt16<true>(v[2]) | ((unsigned)cvt16<true>(v[3]) << 16);
        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(e.out) + o) = pk;
    } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(e.out) + o) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

int gemm_splitk_factor(int dtype, int M, int N, int K, const Epilogue& e) {
    // exact-fp32 GEMMs of one or two utterances: few 64x64 tiles, many K-steps of 32.
    // NOT for the 16-bit modes, although a single utterance's 49 x 384 x 1536 pw2 would gain from it (10.3 -> ~7 us, measured:
    // single-utterance latency 5.1 -> 4.96 ms): there the engine keeps a row's result independent of how many rows the launch
    // has — a split changes the fp32 summation order with M, and a packed batch would stop being bit-identical to the padded one
    // (tests/test_gpu_packed.py::test_trimmed_dense_vocoder_is_bit_identical).
    // (fp32 itself: up to 512 rows and from K = 384 on — a single utterance's 49 x 1536 x 384 pw1 is 24 workgroups x 12 steps, its
    // vocoder's 294 x 512 x 2048 pw2 40 workgroups x 64 steps; 20 -> 12 us and 99 -> 40 us)
    if (dtype != F32 || M > 512 || K < 384 || K % 32 || N % 8 || e.ldo % 4 || e.mode > EPI_RESID) return 1;
    const int tiles = ((M + 63) / 64) * ((N + 63) / 64), nk = K / 32;
    int sk = 1;
    for (int c : {8, 6, 4, 3, 2}) if (nk % c == 0 && nk / c >= 4 && tiles * c <= 256) { sk = c; break; }
    return sk;
}

void launch_gemm_splitk(hipStream_t s, int dtype, const void* A, int lda, const void* W, int ldw, int M, int N, int K, const Epilogue& e,
                        int S, float* workspace /* [S][M][N] */) {
    Epilogue ep;  // partial sums: plain fp32 store, no bias, no mask
    ep.mode = EPI_STORE; ep.out_dtype = F32; ep.out = workspace; ep.ldo = N; ep.ksplit = S;
    launch_gemm(s, dtype, A, lda, W, ldw, M, N, K, ep);
    const int64_t n4 = (int64_t)M * (N / 4);
    const dim3 grid((unsigned)((n4 + 255) / 256));
    if (e.mode == EPI_RESID) STN_KLAUNCH(splitk_reduce_kernel<EPI_RESID>, grid, dim3(256), 0, s, workspace, S, M, N, e);
    else STN_KLAUNCH(splitk_reduce_kernel<EPI_STORE>, grid, dim3(256), 0, s, workspace, S, M, N, e);
}

}  // namespace stn
