// kernels_xattn_hs.hip — the vector estimator's cross-attention block, HEAD-SPLIT: q projection, rotation, attention and output
// projection of ONE HEAD in one launch, the output projection leaving as a 16-bit per-head partial sum (gfx950, wave64, 16-bit modes).
//
//     part[h][row][:] = Wo[:, h*96 .. +96] . attention_h(q_h = Wq[h*96 .. +96, :] . xn[row] + bq_h,  K_h, V_h)          h = 0..3
//
// The block used to be four launches (fold_ln | q GEMM | attention | output GEMM + residual: 10 + 8 + 14 + 10 us at batch 128, 14 % of
// a batch at single-digit matrix-pipe utilisation) and is now fold_ln + this one.  The sum over the heads, the output bias and the
// residual add are exactly the K4-split fold (kernels_fold.hpp): the NEXT reader of x — every cross-attention block is followed by a
// ConvNeXt block, whose fold_dwconv_ln reads x anyway — adds part[0..3] in head order, so no workgroup ever needs another head's
// result and there are no atomics: a replay is bit-identical.  Stands in for part of the body of vector_est_ort_->Run
// (/root/reference/cpp/helper.cpp:643-647).
//
// Why per head.  The per-utterance forms of round 3 (kernels_xattn.hip, retired) put one workgroup on an utterance and had to pull
// BOTH whole matrices (2 x 295 KB) through one CU's L2 ingest (~25-30 B/clk): that, not arithmetic, made them lose.  A workgroup here
// owns (a pair of utterances, one head): it streams a QUARTER of each matrix (2 x 72 KiB) for up to 8 row tiles, and the four
// workgroups of a pair sit on one XCD (equal blockIdx % 8), so the pair's xn rows are fetched into that L2 once.
//
// Shape.  Every product is computed TRANSPOSED so that a lane is a ROW (a latent frame) from the first MFMA to the last store, and each
// accumulator is the next product's B operand without leaving the registers (CDNA4 accumulator-as-operand: registers 8s..8s+7 of lane
// half hf are k = 16s + 8(j>>2) + 4hf + (j&3), so the A side is read / packed in that k order):
//   q^T[d][r]  = sum_c Wq_h[d][c] xn[r][c]      A = Wq_h fragments (LDS, fragment order = launch_repack_frag), B = the wave's xn rows
//   rotation (text blocks: LARoPE) on the accumulators: the pair (d, d + 48) of a row lives in ONE lane — registers (T0[i], T1[i+8]),
//              (T0[i+8], T2[i]), (T1[i], T2[i+8]), i < 8 — so there is no LDS round trip and no barrier
//   S^T[k][r]  = sum_d K[k][d] q[r][d]          A = K rows (LDS, two 8-byte reads per k-step), B = q^T packed to 16 bits
//   softmax over the registers of a lane (+ one cross-half shuffle)
//   O^T[d][r]  = sum_k V^T[d][k] P[r][k]        A = V^T rows (LDS image transposed while staging), B = the exponentials packed
//   y^T[n][r]  = sum_d Wo[n][h*96+d] O[r][d]    A = Wo_h fragments (LDS, accumulator-operand order = launch_repack_frag_acc), B = O^T packed
//   y leaves through a wave-private LDS image as whole 256-byte row segments.
// A wave owns one 32-row tile (two when the pair has more than four tiles) and never exchanges data with another wave: the four
// barriers of the kernel only hand LDS regions over (Wq in | q projection done -> K/V in | Wo in).
// Rounding follows the four-launch form step by step (q rounded to 16 bits after the bias, rotated and scaled in fp32, rounded again;
// exponentials rounded; O scaled by 1/sum in fp32, rounded), so the two forms differ only by the rounding of the per-head partial sums.
//
// All global -> LDS traffic is register-staged (buffer loads with the hardware range check, then ds_write): hipcc puts a vmcnt(0) in
// front of every LDS access that follows an LDS-DMA, which would serialise the prefetches this kernel lives on — K/V travel under the
// q projection, Wo under the attention.
#include "kernels.hpp"
#include "dev_env.hpp"
#include "kernels_dev.hpp"

#include <algorithm>
#include <stdio.h>
#include <stdlib.h>

namespace stn {
namespace {

constexpr int HS_C = 384, HS_DH = 96, HS_H = 4;
constexpr int HS_WB = HS_DH * HS_C * 2;  // bytes of one head's share of Wq or Wo: 72 KiB
constexpr int HS_XS = 128 + 16;          // bytes per row of an xn chunk image (64 channels + pad: the 16 lanes of a ds_read_b128 group hit 16 slots)
constexpr int HS_XB = 32 * HS_XS;        // one chunk image of a 32-row tile
constexpr int HS_RB = HS_DH * 2;         // bytes per K / V row in LDS: unpadded (V: what the transposed read wants; K: chunk-swizzled)
constexpr int HS_IS = 256 + 16;          // bytes per row of an output image (128 channels + pad)
constexpr int HS_IMG = 4 * 32 * HS_IS;   // the four waves' output images
constexpr int HS_KN = (128 * 12 + 255) / 256;  // 16-byte items of K (or V) per thread and slot (<= 128 keys x 12 chunks)

__host__ __device__ constexpr int hs_slot_bytes(int kc) { return 2 * kc * HS_RB; }
__host__ __device__ constexpr int hs_max(int a, int b) { return a > b ? a : b; }
// LDS: [0, 72 KiB) Wq_h, later Wo_h | behind it the waves' xn chunk images, later the K / V slots, later the four output images
__host__ __device__ constexpr int hs_bias_offset(int U, int kc) { return HS_WB + hs_max(hs_max(U * hs_slot_bytes(kc), HS_IMG), 4 * 2 * HS_XB); }
__host__ __device__ constexpr int hs_lds_bytes(int U, int kc) { return hs_bias_offset(U, kc) + HS_DH * 4; }  // ... | bq_h (96 floats)

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef short v4s_t __attribute__((ext_vector_type(4)));
template <bool F16>
__device__ __forceinline__ unsigned pk2(float a, float b) {  // two fp32 -> one 16-bit pair, round to nearest even
    if constexpr (F16) { const f16x2_t h = {(_Float16)a, (_Float16)b}; return __builtin_bit_cast(unsigned, h); }
    else { const bf16x2_t h = {(__bf16)a, (__bf16)b}; return __builtin_bit_cast(unsigned, h); }
}
template <bool F16>
__device__ __forceinline__ void r16x2(float& a, float& b) {  // a pair rounded to the storage format and back
    const unsigned w = pk2<F16>(a, b);
    if constexpr (F16) { const f16x2_t h = __builtin_bit_cast(f16x2_t, w); a = (float)h[0]; b = (float)h[1]; }
    else { a = __uint_as_float(w << 16); b = __uint_as_float(w & 0xFFFF0000u); }
}
template <bool F16>
__device__ __forceinline__ bf16x8 pk8(const float* v) {
    const u32x4 w = {pk2<F16>(v[0], v[1]), pk2<F16>(v[2], v[3]), pk2<F16>(v[4], v[5]), pk2<F16>(v[6], v[7])};
    return __builtin_bit_cast(bf16x8, w);
}
__device__ __forceinline__ u32x4 ldb128(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ void zero16(f32x16& a) {
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = 0.f;
}
// hardware transpose read (gfx950 ds_read_b64_tr_b16): per group of 16 lanes, lane 4q + p supplies the address of row q, columns 4p .. 4p+3
// of a 4 x 16 block of 16-bit values, and lane i receives column i of the four rows
__device__ __forceinline__ uint2 ld_tr(const unsigned char* a) {
    return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(a)));
}

}  // namespace

struct HsArgs {
    const uint16_t* xn; const uint16_t* wq; const float* bq; const uint16_t* kp; const uint16_t* vp; const uint16_t* wo;
    uint16_t* part; int64_t part_stride; int64_t M;
    int ldk, B, G, Gpad, Lk, kc;
    const int* qlen; const int* klen; const int* q_off; const int* k_off;
    const int* pairs;  // U = 2: the two utterances of group g are pairs[2g], pairs[2g+1] (-1: none); null: 2g and 2g + 1
    int rope_mode; float log_base, gamma;
    unsigned long long* ts;  // diagnostics: 8 shader-clock stamps per workgroup
};

namespace {

// U = utterances per workgroup (2 at batch size: one round of workgroups on 256 CUs; 1 below).  Tiles of the workgroup: slot 0's
// ceil(nq/32) tiles, then slot 1's; wave w owns tiles w and w + 4.
template <bool F16, int U>
__global__ __launch_bounds__(256, 1) void xattn_hs_kernel(HsArgs p) {
    constexpr int C = HS_C, DH = HS_DH;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = (int)blockIdx.x % p.Gpad, h = (int)blockIdx.x / p.Gpad;  // the heads of a group share blockIdx % 8: one XCD
    if (g >= p.G) return;
    int nq[U], nk[U], ntl[U];
    int64_t row0[U], krow0[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int b = (U > 1 && p.pairs) ? p.pairs[2 * g + u] : g * U + u;
        const bool ok = b >= 0 && b < p.B;
        const int bb = ok ? b : 0;
        nq[u] = ok ? p.qlen[bb] : 0;
        nk[u] = ok ? (p.klen ? min(p.klen[bb], p.Lk) : p.Lk) : 0;
        row0[u] = p.q_off[bb];
        krow0[u] = p.k_off ? (int64_t)p.k_off[bb] : (int64_t)bb * p.Lk;
        ntl[u] = (nq[u] + 31) >> 5;
    }
    // The rows of the group's utterances, slot 0's then slot 1's, are ONE virtual sequence of N rows cut into 32-row tiles: a pair of
    // 70 + 50 frames is 4 tiles, not 3 + 2, and with the pairs chosen longest-with-shortest (launch_xattn_hs_pairs) every workgroup of a
    // batch of like lengths gets the same number — the launch is as long as its slowest workgroup.  A tile that straddles the two
    // utterances runs the attention against both contexts and keeps, per lane, its own.
    const int nq0 = nq[0], N = nq0 + (U > 1 ? nq[U - 1] : 0);
    const int T = (N + 31) >> 5;
    if (T == 0) return;  // uniform
    unsigned long long t0 = 0;
    if (p.ts) t0 = __builtin_readcyclecounter();
    const int kc = p.kc, SLOT = hs_slot_bytes(kc);
    unsigned char* const WQ = lds;                                  // until the q projection is done
    unsigned char* const WO = lds;                                  // behind it
    unsigned char* const XB = lds + HS_WB + wave * (2 * HS_XB);     // the wave's two xn chunk images (q projection)
    unsigned char* const KV = lds + HS_WB;                          // slot u: K rows (chunk-swizzled), then V rows (attention)
    unsigned char* const IMG = lds + HS_WB + wave * (32 * HS_IS);   // the wave's output image (output projection)

    // this wave's tiles
    bool tv[2]; int trow[2];  // tile e of this wave: virtual rows trow[e] .. trow[e] + 31
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int ti = wave + 4 * e;
        tv[e] = ti < T;
        trow[e] = ti * 32;
    }
    (void)ntl;
    // virtual row -> global row of xn / part (beyond N: none)
    auto grow_of = [&](int v) __attribute__((always_inline)) -> int64_t { return v < nq0 ? row0[0] + v : row0[U - 1] + (v - nq0); };

    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.xn, (size_t)p.M * C * 2);
    const __amdgpu_buffer_rsrc_t rs_wq = make_rsrc(reinterpret_cast<const unsigned char*>(p.wq) + (size_t)h * HS_WB, HS_WB);
    const __amdgpu_buffer_rsrc_t rs_wo = make_rsrc(reinterpret_cast<const unsigned char*>(p.wo) + (size_t)h * HS_WB, HS_WB);
    const __amdgpu_buffer_rsrc_t rs_k = make_rsrc(p.kp, 0x7FFFFFFFu), rs_v = make_rsrc(p.vp, 0x7FFFFFFFu);
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(reinterpret_cast<unsigned char*>(p.part) + (size_t)h * p.part_stride * 2, (size_t)p.M * C * 2);

    // ---- loads: Wq_h (shared: through LDS), then this wave's xn rows in whole 128-byte lines (chunk s = channels [64s, 64s + 64),
    // item = (row, 16-byte piece)) -----------------------------------------------------------------------------------------------------------
    u32x4 ww[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) ww[i] = ldb128(rs_wq, (unsigned)(tid + 256 * i) * 16u);
    auto issue_x = [&](int vrow0, u32x4 (&xw)[24]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int item = j * 64 + lane, r = item >> 3, pc = item & 7;
            const unsigned base = vrow0 + r < N ? (unsigned)(grow_of(vrow0 + r) * (C * 2) + pc * 16) : OOB;
#pragma unroll
            for (int s = 0; s < 6; ++s) xw[s * 4 + j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, base, s * 128, 0));
        }
    };
    u32x4 xw[24];
    if (tv[0]) issue_x(trow[0], xw);
    // the head's q bias through LDS (read back, 16 bytes per register quad, when a tile's accumulators are done)
    float* const BQ = reinterpret_cast<float*>(lds + hs_bias_offset(U, kc));
    if (tid < DH) BQ[tid] = p.bq ? p.bq[h * DH + tid] : 0.f;
    float frq[3][8];  // rotation frequencies of this lane's 24 pairs (computed behind the first tile's MFMAs)
#pragma unroll
    for (int i = 0; i < 18; ++i) *reinterpret_cast<u32x4*>(WQ + (tid + 256 * i) * 16) = ww[i];
    __syncthreads();  // #1: Wq_h is in LDS
    unsigned long long t1 = 0;
    if (p.ts) t1 = __builtin_readcyclecounter();
    // ---- K and V of the slots: in flight under the q projection, committed behind it ---------------------------------------------------------
    u32x4 kw[U][HS_KN], vw[U][HS_KN];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int i = 0; i < HS_KN; ++i) {
            const int idx = tid + 256 * i, key = idx / 12, c = idx - key * 12;
            const unsigned off = (idx < kc * 12 && key < nk[u]) ? (unsigned)(((krow0[u] + key) * p.ldk + h * DH + c * 8) * 2) : OOB;
            kw[u][i] = ldb128(rs_k, off);
            vw[u][i] = ldb128(rs_v, off);
        }

    // ---- q projection + rotation of the wave's tiles -> qB (the B operands of QK^T) ----------------------------------------------------
    const float qmul = rsqrtf((float)DH) * 1.44269504088896340736f;
    bf16x8 qB[2][6];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        if (!tv[e]) continue;  // wave-uniform
        f32x16 qa[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) zero16(qa[t]);
        // Software pipeline by hand (one wave per SIMD: nothing else hides an LDS read): the 4 B and 12 A fragments of chunk s + 1 are
        // read — behind the ds_writes of its xn image — before the 12 MFMAs of chunk s are issued.
        bf16x8 fb[2][4], fa[2][12];
        auto stage_chunk = [&](int s) __attribute__((always_inline)) {
            unsigned char* const xb = XB + (s & 1) * HS_XB;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int item = j * 64 + lane;
                *reinterpret_cast<u32x4*>(xb + (item >> 3) * HS_XS + (item & 7) * 16) = xw[s * 4 + j];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                fb[s & 1][j] = *reinterpret_cast<const bf16x8*>(xb + lr * HS_XS + (2 * j + lh) * 16);
#pragma unroll
                for (int t = 0; t < 3; ++t) fa[s & 1][3 * j + t] = *reinterpret_cast<const bf16x8*>(WQ + ((t * 24 + 4 * s + j) * 64 + lane) * 16);
            }
        };
        stage_chunk(0);
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            if (s + 1 < 6) stage_chunk(s + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 3; ++t) qa[t] = mfma16<F16>(fa[s & 1][3 * j + t], fb[s & 1][j], qa[t]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (e == 0 && tv[1]) issue_x(trow[1], xw);  // the second tile's rows travel under the rotation of the first
        if (e == 0 && p.rope_mode >= 0) {
            // in revolutions per position unit (the attention kernel's inv_rev table, same expression): pair index 16g + dl(i),
            // dl(i) = (i & 3) + 8 (i >> 2) + 4 lh, for register i < 8 and pair group g < 3
#pragma unroll
            for (int gq = 0; gq < 3; ++gq)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int pidx = 16 * gq + (i & 3) + 8 * (i >> 2) + 4 * lh;
                    frq[gq][i] = __expf(-p.log_base * (float)(2 * pidx) / (float)DH) * 0.15915494309189535f;
                }
        }
        float qv[3][16];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {  // q + bias (register 4qd + j of tile t is d = 32t + 8qd + 4lh + j), rounded to 16 bits like the q GEMM's store
                const float4 bv = *reinterpret_cast<const float4*>(BQ + 32 * t + 8 * qd + 4 * lh);
                qv[t][4 * qd + 0] = qa[t][4 * qd + 0] + bv.x; qv[t][4 * qd + 1] = qa[t][4 * qd + 1] + bv.y;
                qv[t][4 * qd + 2] = qa[t][4 * qd + 2] + bv.z; qv[t][4 * qd + 3] = qa[t][4 * qd + 3] + bv.w;
                r16x2<F16>(qv[t][4 * qd + 0], qv[t][4 * qd + 1]);
                r16x2<F16>(qv[t][4 * qd + 2], qv[t][4 * qd + 3]);
            }
        if (p.rope_mode >= 0) {
            const int v = trow[e] + lr;                    // this lane's virtual row: its utterance, and its position inside it
            const bool in1 = U > 1 && v >= nq0;
            const int nqu = in1 ? nq[U - 1] : nq0;
            const float pscale = p.rope_mode == 1 ? p.gamma / (float)(nqu > 0 ? nqu : 1) : 1.f;
            const float pp = (float)(in1 ? v - nq0 : v) * pscale;
            auto rot = [&](float& a0, float& a1, float fr) __attribute__((always_inline)) {
                const float rev = __builtin_amdgcn_fractf(pp * fr);
                const float sn = __builtin_amdgcn_sinf(rev), cs = __builtin_amdgcn_cosf(rev);
                const float y0 = (a0 * cs - a1 * sn) * qmul, y1 = (a1 * cs + a0 * sn) * qmul;
                a0 = y0; a1 = y1;
            };
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                rot(qv[0][i], qv[1][i + 8], frq[0][i]);       // d = dl         with d + 48 = 32 + (16 + dl)
                rot(qv[0][i + 8], qv[2][i], frq[1][i]);       // d = 16 + dl    with d + 48 = 64 + dl
                rot(qv[1][i], qv[2][i + 8], frq[2][i]);       // d = 32 + dl    with d + 48 = 64 + (16 + dl)
            }
        } else {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) qv[t][i] *= qmul;
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) { qB[e][2 * t] = pk8<F16>(&qv[t][0]); qB[e][2 * t + 1] = pk8<F16>(&qv[t][8]); }
    }
    __syncthreads();  // #2: every wave is done with Wq_h and its chunk images
    unsigned long long t2 = 0;
    if (p.ts) t2 = __builtin_readcyclecounter();

    // ---- K and V rows into their slots: 16-byte copies (V as it is: the transposed read does the rest; K with the 16-byte chunks of
    // row r XOR-ed by (r >> 2) & 3 inside each 64-byte group, which spreads the 32 rows of a fragment read over the banks) -----------------
#pragma unroll
    for (int u = 0; u < U; ++u) {
        unsigned char* const Kb = KV + u * SLOT;
        unsigned char* const Vb = Kb + kc * HS_RB;
#pragma unroll
        for (int i = 0; i < HS_KN; ++i) {
            const int idx = tid + 256 * i, key = idx / 12, c = idx - key * 12;
            if (idx < kc * 12) {
                *reinterpret_cast<u32x4*>(Kb + key * HS_RB + ((c ^ ((key >> 2) & 3)) << 4)) = kw[u][i];
                *reinterpret_cast<u32x4*>(Vb + idx * 16) = vw[u][i];
            }
        }
    }
    // Wo_h: in flight under the attention
#pragma unroll
    for (int i = 0; i < 18; ++i) ww[i] = ldb128(rs_wo, (unsigned)(tid + 256 * i) * 16u);
    __syncthreads();  // #3: K and V of the slots are in LDS
    unsigned long long t3 = 0;
    if (p.ts) t3 = __builtin_readcyclecounter();

    // ---- attention of the wave's tiles -> oB (the B operands of the output projection) ---------------------------------------------------
    const int nkt = kc >> 5;
    const int ksw = ((lr >> 2) & 3) << 4;                                // this lane's K row swizzle
    const int vtr = ((lr & 15) >> 2) * HS_RB + ((lr >> 4) * 16 + (lr & 3) * 4) * 2;  // its address inside a transposed-read block
    bf16x8 oB[2][6];
    // S^T of one 32-row tile against slot `slot`'s keys: all six K fragments of a key tile are read before its MFMAs
    auto scores = [&](int slot, const bf16x8 (&q)[6], f32x16 (&sc)[4]) __attribute__((always_inline)) {
        const unsigned char* const Kb = KV + slot * SLOT;
        // this lane's four swizzled chunk positions inside a 64-byte group of its K row: chunk c sits at kq[c & 3] + 64 (c >> 2)
        const unsigned char* kq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) kq[j] = Kb + lr * HS_RB + 8 * lh + ((j << 4) ^ ksw);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt < nkt) {
                zero16(sc[kt]);
                u32x4 ka[6];
#pragma unroll
                for (int ks = 0; ks < 6; ++ks) {  // d = 16ks + 4lh + {0..3} and + 8: the two 8-byte halves of chunks 2ks and 2ks + 1
                    const uint2 lo = *reinterpret_cast<const uint2*>(kq[(2 * ks) & 3] + kt * 32 * HS_RB + ((2 * ks) >> 2) * 64);
                    const uint2 hi = *reinterpret_cast<const uint2*>(kq[(2 * ks + 1) & 3] + kt * 32 * HS_RB + ((2 * ks + 1) >> 2) * 64);
                    ka[ks] = u32x4{lo.x, lo.y, hi.x, hi.y};
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 6; ++ks) sc[kt] = mfma16<F16>(__builtin_bit_cast(bf16x8, ka[ks]), q[ks], sc[kt]);
            }
        }
    };
    // O^T += V_slot^T P^T for one key tile: V^T[d = 32nd + lr][keys 16sidx + 4lh + {0..3} and + 8] by two transposed reads of 4 keys x 16
    // dims per operand, all six operands of the key tile read before its MFMAs
    auto pv_tile = [&](int slot, int kt, const float (&pr)[16], f32x16 (&oa)[3]) __attribute__((always_inline)) {
        const unsigned char* const Vb = KV + slot * SLOT + kc * HS_RB;
        u32x4 va[2][3];
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
            const unsigned char* const vb = Vb + (kt * 32 + 16 * sidx + 4 * lh) * HS_RB + vtr;
#pragma unroll
            for (int nd = 0; nd < 3; ++nd) {
                const uint2 lo = ld_tr(vb + nd * 64), hi = ld_tr(vb + nd * 64 + 8 * HS_RB);
                va[sidx][nd] = u32x4{lo.x, lo.y, hi.x, hi.y};
            }
        }
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
            const bf16x8 pb = pk8<F16>(&pr[8 * sidx]);
#pragma unroll
            for (int nd = 0; nd < 3; ++nd) oa[nd] = mfma16<F16>(__builtin_bit_cast(bf16x8, va[sidx][nd]), pb, oa[nd]);
        }
    };
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        if (!tv[e]) continue;
        // which contexts the tile's rows belong to (wave-uniform), and this lane's own
        const bool has0 = trow[e] < nq0, has1 = U > 1 && min(trow[e] + 32, N) > nq0;
        const bool both = has0 && has1;
        const int s0 = has0 ? 0 : 1;                     // the (first) slot of the tile
        const bool in1 = U > 1 && trow[e] + lr >= nq0;   // this lane's row is slot 1's
        const int nku = in1 ? nk[U - 1] : nk[0];         // its context length
        const int nk_s0 = s0 ? nk[U - 1] : nk[0];
        f32x16 sc[4];
        scores(s0, qB[e], sc);
        if (both) {  // a tile that straddles the two utterances: the other context too, each lane keeps its own scores
            f32x16 sc1[4];
            scores(1, qB[e], sc1);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
                if (kt < nkt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) sc[kt][i] = in1 ? sc1[kt][i] : sc[kt][i];
        }
        // masked maximum, exponentials and their sum: element i of key tile kt is key 32kt + (i & 3) + 8 (i >> 2) + 4 lh; only a key
        // tile that holds the context's end needs the comparison (wave-uniform; a straddling tile always compares)
        float m = -1e30f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
            if (kt < nkt) {
                if (!both && kt * 32 + 32 <= nk_s0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) m = fmaxf(m, sc[kt][i]);
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) m = fmaxf(m, kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh < nku ? sc[kt][i] : -1e30f);
                }
            }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float lsum = 0.f;
        f32x16 oa[3];
#pragma unroll
        for (int nd = 0; nd < 3; ++nd) zero16(oa[nd]);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
            if (kt < nkt) {
                float pr[16];
                // (v_exp_f32 flushes results below 2^-126 to zero where exp2f returns a denormal: the same 16-bit value, and a sum that holds
                // the maximum's 1.0 does not see the difference)
                if (!both && kt * 32 + 32 <= nk_s0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) { pr[i] = __builtin_amdgcn_exp2f(sc[kt][i] - m); lsum += pr[i]; }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        pr[i] = kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh < nku ? __builtin_amdgcn_exp2f(sc[kt][i] - m) : 0.f;
                        lsum += pr[i];
                    }
                }
                if (!both) {
                    pv_tile(s0, kt, pr, oa);
                } else {  // each context's values weigh only its own lanes' exponentials
                    float p0[16], p1[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) { p0[i] = in1 ? 0.f : pr[i]; p1[i] = in1 ? pr[i] : 0.f; }
                    pv_tile(0, kt, p0, oa);
                    pv_tile(1, kt, p1, oa);
                }
            }
        lsum += __shfl_xor(lsum, 32, 64);
        const float inv = (nku > 0 && lsum > 0.f) ? 1.0f / lsum : 0.f;
#pragma unroll
        for (int nd = 0; nd < 3; ++nd) {
            float ov[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) ov[i] = oa[nd][i] * inv;
            oB[e][2 * nd] = pk8<F16>(&ov[0]);
            oB[e][2 * nd + 1] = pk8<F16>(&ov[8]);
        }
    }
    unsigned long long t4 = 0;
    if (p.ts) t4 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 18; ++i) *reinterpret_cast<u32x4*>(WO + (tid + 256 * i) * 16) = ww[i];
    __syncthreads();  // #4: Wo_h is in LDS; every wave is done with K / V (the output images overlay them)
    unsigned long long t5 = 0;
    if (p.ts) t5 = __builtin_readcyclecounter();

    // ---- output projection: all 12 channel tiles of a row tile, then out through the wave's image, 128 channels (whole 256-byte row
    // segments) at a time; rows past the utterance's end go nowhere (store offsets beyond the buffer) ---------------------------------------
    const int srow = lane >> 4, scol = (lane & 15) * 16;  // this lane's piece of a store instruction: 4 rows x 256 bytes
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        if (!tv[e]) continue;
        f32x16 ya[12];
#pragma unroll
        for (int n = 0; n < 12; ++n) zero16(ya[n]);
        bf16x8 wa[2][12];
#pragma unroll
        for (int n = 0; n < 12; ++n) wa[0][n] = *reinterpret_cast<const bf16x8*>(WO + ((n * 2) * 64 + lane) * 16);
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            if (ks + 1 < 6) {
#pragma unroll
                for (int n = 0; n < 12; ++n)
                    wa[(ks + 1) & 1][n] = *reinterpret_cast<const bf16x8*>(WO + (((((ks + 1) >> 1) * 12 + n) * 2 + ((ks + 1) & 1)) * 64 + lane) * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < 12; ++n) ya[n] = mfma16<F16>(wa[ks & 1][n], oB[e][ks], ya[n]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq)
                    *reinterpret_cast<uint2*>(IMG + lr * HS_IS + (32 * n + 8 * qq + 4 * lh) * 2) =
                        make_uint2(pk2<F16>(ya[4 * c + n][4 * qq], ya[4 * c + n][4 * qq + 1]), pk2<F16>(ya[4 * c + n][4 * qq + 2], ya[4 * c + n][4 * qq + 3]));
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int row = 4 * k + srow;
                const u32x4 v = *reinterpret_cast<const u32x4*>(IMG + row * HS_IS + scol);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_o, trow[e] + row < N ? (unsigned)(grow_of(trow[e] + row) * (C * 2) + c * 256 + scol) : OOB, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (p.ts && tid == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        unsigned long long* tp = p.ts + (size_t)blockIdx.x * 8;
        tp[0] = t0; tp[1] = t1; tp[2] = t2; tp[3] = t3; tp[4] = t4; tp[5] = t5; tp[6] = __builtin_readcyclecounter(); tp[7] = (unsigned long long)T;
    }
}

// Pairs for the head-split launch: utterances sorted by length (ties: by index), the k-th longest paired with the k-th shortest — the
// sums of a pair's lengths, hence its row tiles, are as equal as the batch allows.  pairs[2g], pairs[2g + 1] (-1: the odd one out).
__global__ __launch_bounds__(1024) void hs_pairs_kernel(const int* __restrict__ len, int B, int* __restrict__ pairs) {
    __shared__ int l[1024], sorted[1024];
    const int i = threadIdx.x;
    if (i < B) l[i] = len[i];
    __syncthreads();
    if (i < B) {
        int rank = 0;
        for (int j = 0; j < B; ++j) rank += (l[j] < l[i] || (l[j] == l[i] && j < i)) ? 1 : 0;
        sorted[rank] = i;  // ascending
    }
    __syncthreads();
    const int G = (B + 1) / 2;
    if (i < G) {
        pairs[2 * i] = sorted[B - 1 - i];
        pairs[2 * i + 1] = B - 1 - i > i ? sorted[i] : -1;
    }
}

template <bool F16, int U>
void hs_launch(hipStream_t s, const HsArgs& a, size_t lds) {
    static PerDeviceOnce attr_once;
    if (attr_once.need())
        stn_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_hs_kernel<F16, U>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                      "hipFuncSetAttribute(xattn_hs)");
    STN_KLAUNCH((xattn_hs_kernel<F16, U>), dim3((unsigned)(a.Gpad * HS_H)), dim3(256), lds, s, a);
}

// utterances per workgroup: two when one per workgroup would need more than one round of workgroups on the chip (and the pair's K / V
// slots fit beside Wo_h); a pair's tiles must fit the two a wave can own (<= 128 rows per utterance)
int hs_group(int B, int L, int kc) {
    static const int force = [] { const char* e = stn::dev_env("STN_XATTN_HS_U"); return e ? atoi(e) : 0; }();  // A/B switch
    const bool can2 = L <= 128 && (size_t)hs_lds_bytes(2, kc) <= 160 * 1024;
    if (force == 1 || !can2) return 1;
    if (force == 2) return 2;
    return B * HS_H > 256 ? 2 : 1;
}

}  // namespace

int xattn_hs_group(int B, int L, int Lk) { return hs_group(B, L, (Lk + 31) & ~31); }
void launch_xattn_hs_pairs(hipStream_t s, const int* qlen, int B, int* pairs) {
    if (B < 1 || B > 1024) throw std::invalid_argument("launch_xattn_hs_pairs: 1 <= B <= 1024");
    STN_KLAUNCH(hs_pairs_kernel, dim3(1), dim3(1024), 0, s, qlen, B, pairs);
}

bool xattn_hs_supported(int dtype, int C, int H, int L, int Lk, int ldk) {
    return is_half(dtype) && C == HS_C && H == HS_H && L >= 1 && L <= 256 && Lk >= 1 && Lk <= 128 && ldk % 8 == 0;
}

void launch_xattn_hs(hipStream_t s, int dtype, const void* xn, int64_t M, const void* WqF, const float* bq, const void* kp, const void* vp, int ldk,
                     const void* WoA, void* part, int64_t part_stride, int B, int L, int Lk, const int* qlen, const int* klen,
                     const int* q_off, const int* k_off, int rope_mode, float rope_base, float rope_gamma, unsigned long long* ts, const int* pairs) {
    if (B == 0 || L == 0 || M == 0) return;
    if (!xattn_hs_supported(dtype, HS_C, HS_H, L, Lk, ldk) || !qlen || !q_off || (k_off && !klen) || !part || part_stride < M * HS_C ||
        M * HS_C * 2 >= 0x7FFFFFFFll || (int64_t)B * Lk * ldk * 2 >= 0x7FFFFFFFll ||
        ((reinterpret_cast<uintptr_t>(xn) | reinterpret_cast<uintptr_t>(WqF) | reinterpret_cast<uintptr_t>(WoA) | reinterpret_cast<uintptr_t>(kp) |
          reinterpret_cast<uintptr_t>(vp) | reinterpret_cast<uintptr_t>(part)) & 15) || (bq && (reinterpret_cast<uintptr_t>(bq) & 15)))
        throw std::invalid_argument("launch_xattn_hs: unsupported shape, missing packed-row maps or misaligned operands (callers check xattn_hs_supported)");
    HsArgs a;
    a.xn = static_cast<const uint16_t*>(xn); a.wq = static_cast<const uint16_t*>(WqF); a.bq = bq;
    a.kp = static_cast<const uint16_t*>(kp); a.vp = static_cast<const uint16_t*>(vp); a.wo = static_cast<const uint16_t*>(WoA);
    a.part = static_cast<uint16_t*>(part); a.part_stride = part_stride; a.M = M; a.ldk = ldk; a.B = B; a.Lk = Lk;
    a.kc = (Lk + 31) & ~31;
    const int U = hs_group(B, L, a.kc);
    a.G = (B + U - 1) / U; a.Gpad = (a.G + 7) & ~7;
    a.qlen = qlen; a.klen = klen; a.q_off = q_off; a.k_off = k_off; a.pairs = pairs;
    a.rope_mode = rope_mode; a.log_base = logf(rope_base); a.gamma = rope_gamma; a.ts = ts;
    const size_t lds = (size_t)hs_lds_bytes(U, a.kc);
    if (dtype == F16) { if (U == 2) hs_launch<true, 2>(s, a, lds); else hs_launch<true, 1>(s, a, lds); }
    else { if (U == 2) hs_launch<false, 2>(s, a, lds); else hs_launch<false, 1>(s, a, lds); }
}

}  // namespace stn
