// kernels_fold.hpp — device helpers of the K4-split fold (kernels_misc.hip: fold_ln / fold_dwconv_ln; kernels_xattn_hs.hip feeds the same fold): ONE
// definition of the summation order, so that every reader of a pending update computes the same bits.
//   x_new = x + gamma * ((((p0 + p1) + p2) + ... + p_{S-1}) + b2) + rowvec[seq]          (fp32)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace stn {

// Splits a fold accepts: ffn_split_choose's values (kernels_ffn.hip).  The partial sums are added in split order, one after the other.
template <bool F16>
__device__ __forceinline__ float p16_to_f(unsigned h) {  // one 16-bit partial (low 16 bits of h) -> fp32
    if constexpr (F16) { const _Float16 v = __builtin_bit_cast(_Float16, (uint16_t)h); return (float)v; }
    else return __uint_as_float(h << 16);
}
// acc[0..N) (+)= the N consecutive channels at `p16` of splits [s0, s0 + CH): CH loads of N * 2 bytes issued together, then added
// in split order.  FIRST: acc starts from split s0 instead of being added to.
template <bool F16, int N, int CH, bool FIRST>
__device__ __forceinline__ void fold_chunk(const uint16_t* __restrict__ p16, int64_t pstride, int s0, float (&acc)[N]) {
    unsigned w[CH][N / 2];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const uint16_t* q = p16 + (size_t)(s0 + c) * pstride;
        if constexpr (N == 8) { const uint4 u = *reinterpret_cast<const uint4*>(q); w[c][0] = u.x; w[c][1] = u.y; w[c][2] = u.z; w[c][3] = u.w; }
        else { const uint2 u = *reinterpret_cast<const uint2*>(q); w[c][0] = u.x; w[c][1] = u.y; }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int j = 0; j < N / 2; ++j) {
            const float lo = p16_to_f<F16>(w[c][j] & 0xFFFFu), hi = p16_to_f<F16>(w[c][j] >> 16);
            if (FIRST && c == 0) { acc[2 * j] = lo; acc[2 * j + 1] = hi; }
            else { acc[2 * j] += lo; acc[2 * j + 1] += hi; }
        }
}
// the whole sum over S splits, in chunks of at most 12 loads in flight
template <bool F16, int N, int S>
__device__ __forceinline__ void fold_sum(const uint16_t* __restrict__ p16, int64_t pstride, float (&acc)[N]) {
    constexpr int CH = S <= 12 ? S : 12;
    static_assert(S % CH == 0, "split count");
    fold_chunk<F16, N, CH, true>(p16, pstride, 0, acc);
#pragma unroll
    for (int s0 = CH; s0 < S; s0 += CH) fold_chunk<F16, N, CH, false>(p16, pstride, s0, acc);
}
__device__ __forceinline__ float fold_one(float x, float y, float b2, float gm, float rv) { return x + gm * (y + b2) + rv; }
__device__ __forceinline__ float4 fold_four(float4 x, const float* acc, float4 b2, float4 gm, float4 rv) {
    return make_float4(fold_one(x.x, acc[0], b2.x, gm.x, rv.x), fold_one(x.y, acc[1], b2.y, gm.y, rv.y), fold_one(x.z, acc[2], b2.z, gm.z, rv.z),
                       fold_one(x.w, acc[3], b2.w, gm.w, rv.w));
}


// sum over each half of the wavefront (lanes 0-31 / 32-63), result in every lane of the half: four DPP steps inside a row of 16
// (quad swaps, half-row mirror, row mirror) and one swizzle across the two rows — no LDS-crossbar round trip per step
__device__ __forceinline__ float dpp_add(float v, int ctrl_sel) {
    const int iv = __float_as_int(v);
    int o;
    switch (ctrl_sel) {
        case 0: o = __builtin_amdgcn_update_dpp(0, iv, 0xB1, 0xF, 0xF, true); break;   // quad_perm [1,0,3,2]
        case 1: o = __builtin_amdgcn_update_dpp(0, iv, 0x4E, 0xF, 0xF, true); break;   // quad_perm [2,3,0,1]
        case 2: o = __builtin_amdgcn_update_dpp(0, iv, 0x141, 0xF, 0xF, true); break;  // row_half_mirror
        default: o = __builtin_amdgcn_update_dpp(0, iv, 0x140, 0xF, 0xF, true); break; // row_mirror
    }
    return v + __int_as_float(o);
}
__device__ __forceinline__ float half_wave_sum(float v) {
    v = dpp_add(v, 0); v = dpp_add(v, 1); v = dpp_add(v, 2); v = dpp_add(v, 3);
    return v + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));  // lane ^ 16 inside each group of 32
}


}  // namespace stn
