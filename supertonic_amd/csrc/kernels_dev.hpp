// kernels_dev.hpp — device-side helpers shared by the MFMA kernels (kernels_gemm.hip, kernels_ffn.hip): vector types,
// buffer descriptors, LDS-DMA, counted waits, the 16-bit conversions and the GELU forms.  One definition, so that a fused
// kernel and the launches it replaces round identically.
#pragma once
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

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


// erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7): GELU(x) = 0.5 x (1 + erf(x / sqrt2))
__device__ __forceinline__ float gelu_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);  // 1 ulp; the A&S fit itself is 1.5e-7
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * __expf(-z * z);
    const float erf_v = x < 0.f ? -erf_abs : erf_abs;
    return 0.5f * x * (1.0f + erf_v);
}
// the tanh form of GELU, 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))) = x sigmoid(2 sqrt(2/pi) x (1 + 0.044715 x^2)), in full fp32
// precision: what a graph that spells GELU with Tanh (or Gelu approximate="tanh") computes (fp32 / f16 outputs; bf16 outputs take
// gelu_bf16_f below, the same function through v_exp / v_rcp)
__device__ __forceinline__ float gelu_tanh_f(float x) {
    return x / (1.0f + expf(-1.5957691216057308f * x * (1.0f + 0.044715f * x * x)));
}
__device__ __forceinline__ float act_f(float v, int act) {
    if (act == ACT_GELU) return gelu_f(v);
    if (act == ACT_GELU_TANH) return gelu_tanh_f(v);
    if (act == ACT_SILU) return v / (1.0f + expf(-v));
    return v;
}
// GELU for results that are about to be rounded to bf16 (8 significant bits): x * sigmoid(1.59577 x (1 + 0.044715 x^2)),
// the tanh form written as a sigmoid — 5 VALU + v_exp_f32 + v_rcp_f32 instead of ~20 + 2.  |error| <= 3e-4 absolute,
// below a bf16 ulp wherever |gelu(x)| > 0.08 and relatively tiny near 0.  fp32 outputs keep the erf form above.
__device__ __forceinline__ float gelu_bf16_f(float x) {
    const float t = x * fmaf(x * x, -0.10294324f, -2.30220819f);  // -(1.5957691 + 0.0713548 x^2) x * log2(e)
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}
__device__ __forceinline__ float act_out_f(float v, int act, bool to_bf16) {
    if ((act == ACT_GELU || act == ACT_GELU_TANH) && to_bf16) return gelu_bf16_f(v);
    return act_f(v, act);
}
// bias + activation + row mask on an 8-column group; the activation kind is resolved ONCE per group (a per-element
// switch compiles to scalar branches around every element and triples the epilogue's issue time)
__device__ __forceinline__ void act8(float (&v)[8], const float (&bias)[8], int act, bool to_bf16, float keep) {
    if (act == ACT_NONE) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (v[j] + bias[j]) * keep;
    } else if ((act == ACT_GELU || act == ACT_GELU_TANH) && to_bf16) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = gelu_bf16_f(v[j] + bias[j]) * keep;
    } else if (act == ACT_GELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = gelu_f(v[j] + bias[j]) * keep;
    } else if (act == ACT_GELU_TANH) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = gelu_tanh_f(v[j] + bias[j]) * keep;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float t = v[j] + bias[j]; v[j] = t / (1.0f + expf(-t)) * keep; }
    }
}
__device__ __forceinline__ uint16_t f2bf(float f) {
    __hip_bfloat16 h = __float2bfloat16(f);
    return *reinterpret_cast<uint16_t*>(&h);
}
// 16-bit formats: bf16 (F16 = false) or IEEE half (F16 = true, the STN_DTYPE_F16 mode): same tiles, same LDS images, same
// MFMA timing (v_mfma_f32_32x32x16_f16); only the conversion and the instruction differ, both resolved at compile time.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool F16>
__device__ __forceinline__ uint16_t cvt16(float f) {
    if constexpr (F16) { const _Float16 h = (_Float16)f; return __builtin_bit_cast(uint16_t, h); }  // v_cvt_f16_f32, RNE
    else return f2bf(f);
}
template <bool F16>
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

#define STN_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// LDS-DMA of one 1-KiB piece (16 B per lane).  Kept in a non-template function: with value-dependent arguments
// the amdgcn builtin is re-checked at template instantiation on the HOST pass, fails there, and the failure is
// swallowed as a substitution failure (the kernel silently loses its host stub).
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned char* lds_dst, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, STN_LDS_PTR(lds_dst), 16, voff, soff, 0, 0);
}

template <int N_>
__device__ __forceinline__ void wait_vm() {
    static_assert(N_ >= 0 && N_ < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

template <int PER, int D>  // wait until at most min(ahead, D) * PER of this wave's DMAs are outstanding
__device__ __forceinline__ void wait_stage(int ahead) {
    if constexpr (D >= 7) { if (ahead >= 7) { wait_vm<7 * PER>(); return; } }
    if constexpr (D >= 6) { if (ahead >= 6) { wait_vm<6 * PER>(); return; } }
    if constexpr (D >= 5) { if (ahead >= 5) { wait_vm<5 * PER>(); return; } }
    if constexpr (D >= 4) { if (ahead >= 4) { wait_vm<4 * PER>(); return; } }
    if constexpr (D >= 3) { if (ahead >= 3) { wait_vm<3 * PER>(); return; } }
    if constexpr (D >= 2) { if (ahead >= 2) { wait_vm<2 * PER>(); return; } }
    if constexpr (D >= 1) { if (ahead >= 1) { wait_vm<1 * PER>(); return; } }
    wait_vm<0>();
}

}  // namespace stn
