// pk_probe — does a packed-fp32 VALU sequence lose results when a co-resident wave does something else? (gfx950)
// Victim phase: the Q-staging arithmetic of kernels_attn.hip (bf16 pairs -> fp32 -> rotate by (cs,sn)=(1,0) -> scale -> bf16),
// compared lane by lane with a scalar restatement that the compiler cannot pack.  Aggressor phase: MFMA / LDS / VALU / sleep.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    const unsigned ra = (ua + 0x7FFFu + ((ua >> 16) & 1u)) >> 16, rb = (ub + 0x7FFFu + ((ub >> 16) & 1u)) >> 16;
    return ra | (rb << 16);
}
__device__ __forceinline__ float mul_nopack(float a, float b) {
    float r;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__global__ __launch_bounds__(256) void probe(const uint32_t* __restrict__ in, int nvec, unsigned* __restrict__ stats,
                                             int mode, int reps, int rope_mode, float mul, float pscale, int phases, int asm_victim) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ float inv_rev[32];
    const int tid = threadIdx.x;
    if (rope_mode >= 0) { for (int i = tid; i < 32; i += 256) inv_rev[i] = __expf(-9.2f * (float)(2 * i) / 64.f) * 0.159f; __syncthreads(); }
    const bool rot = rope_mode >= 0;
    unsigned bad = 0, bad_hi_lanes = 0;
    float sink = 0.f;
    for (int ph = 0; ph < phases; ++ph) {
        const bool victim = ((blockIdx.x + ph) & 1) == 0;
        if (victim && asm_victim) {
            for (int rep = 0; rep < reps; ++rep) {
                const int idx = (int)(((unsigned)blockIdx.x * 977u + (unsigned)rep * 256u + (unsigned)tid) % (unsigned)nvec);
                const u32x4_t w0 = *reinterpret_cast<const u32x4_t*>(in + (size_t)idx * 8);
                const u32x4_t w1 = *reinterpret_cast<const u32x4_t*>(in + (size_t)idx * 8 + 4);
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    float y0, y1;
#define PRE "v_mov_b32 v40, %2\n\tv_mov_b32 v41, %3\n\tv_mov_b32 v44, 1.0\n\tv_mov_b32 v45, 0\n\tv_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\tv_mul_f32 v49, v44, v44\n\tv_and_b32 v47, 0xffff0000, v40\n\tv_and_b32 v46, 0xffff0000, v41\n\tv_mul_f32 v50, v44, v49\n\t"
#define PK1 "v_pk_mul_f32 v[48:49], v[44:45], v[46:47] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
#define PK2 "v_pk_mul_f32 v[44:45], v[44:45], v[46:47]\n\t"
#define POST "v_sub_f32 %0, v48, v49\n\tv_add_f32 %1, v44, v45\n\t"
#define OPS : "=v"(y0), "=v"(y1) : "v"(w0[e2]), "v"(w1[e2]) : "v40", "v41", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v52", "v53"
                    if (asm_victim == 1) asm volatile(PRE PK1 PK2 POST OPS);                                    // as generated
                    else if (asm_victim == 2) asm volatile(PRE PK1 "s_nop 0\n\t" PK2 POST OPS);                 // gap between the two packed ops
                    else if (asm_victim == 3) asm volatile(PRE PK1 PK2 "s_nop 0\n\t" POST OPS);                 // gap before the consumers
                    else if (asm_victim == 4) asm volatile(PRE PK1 "v_pk_mul_f32 v[52:53], v[44:45], v[46:47]\n\t" "v_sub_f32 %0, v48, v49\n\tv_add_f32 %1, v52, v53\n\t" OPS);  // second op does not overwrite its sources
                    else if (asm_victim == 5) asm volatile(PRE PK1 "v_mul_f32 v52, v44, v46\n\tv_mul_f32 v53, v45, v47\n\t" "v_sub_f32 %0, v48, v49\n\tv_add_f32 %1, v52, v53\n\t" OPS);  // only ONE packed op
                    else if (asm_victim == 6) asm volatile(PRE "v_mul_f32 v48, v44, v47\n\tv_mul_f32 v49, v45, v46\n\t" PK2 POST OPS);   // only the second packed op
                    else if (asm_victim == 7) asm volatile(PRE PK1 PK2 "s_nop 3\n\t" POST OPS);
                    else if (asm_victim == 8) { asm volatile(PRE "v_pk_mul_f32 v[48:49], v[44:45], v[46:47]\n\t" "v_sub_f32 %0, v48, v49\n\tv_mov_b32 %1, v46\n\t" OPS); y0 = __uint_as_float(y0 == __uint_as_float(w1[e2] & 0xFFFF0000u) ? (w0[e2] & 0xFFFF0000u) : 0x7fc00000u); }  // pk1 WITHOUT op_sel
                    else { asm volatile(PRE "v_mov_b32 v52, 0\n\tv_mov_b32 v53, 0\n\t" "v_pk_fma_f32 v[48:49], v[44:45], v[46:47], v[52:53] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t" "v_sub_f32 %0, v48, v49\n\tv_mov_b32 %1, v46\n\t" OPS); }  // packed fma with the same cross selection
                    const bool b0 = __float_as_uint(y0) != (w0[e2] & 0xFFFF0000u), b1 = __float_as_uint(y1) != (w1[e2] & 0xFFFF0000u);
                    if (b0 || b1) { ++bad; if ((tid & 63) >= 48) ++bad_hi_lanes; }
                }
            }
        } else if (victim) {
            for (int rep = 0; rep < reps; ++rep) {
                const int idx = (int)(((unsigned)blockIdx.x * 977u + (unsigned)rep * 256u + (unsigned)tid) % (unsigned)nvec);
                const int pos = idx & 127, c = tid & 3;
                const u32x4_t w0 = *reinterpret_cast<const u32x4_t*>(in + (size_t)idx * 8);
                const u32x4_t w1 = *reinterpret_cast<const u32x4_t*>(in + (size_t)idx * 8 + 4);
                u32x4_t o0, o1;
                if (rot || mul != 1.f) {
                    const float pp = (float)pos * pscale;
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) {
                        float a0[2] = {__uint_as_float(w0[e2] << 16), __uint_as_float(w0[e2] & 0xFFFF0000u)};
                        float a1[2] = {__uint_as_float(w1[e2] << 16), __uint_as_float(w1[e2] & 0xFFFF0000u)};
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
                        o0[e2] = pack_bf16x2(y0[0], y0[1]);
                        o1[e2] = pack_bf16x2(y1[0], y1[1]);
                    }
                } else { o0 = w0; o1 = w1; }
                *reinterpret_cast<u32x4_t*>(lds + tid * 32) = o0;
                *reinterpret_cast<u32x4_t*>(lds + tid * 32 + 16) = o1;
                if (!rot) {  // scalar restatement (cs = 1, sn = 0: y = a * mul exactly, barring signed zeros which cannot differ after bf16 packing of +-0 ... compare magnitudes)
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) {
                        const unsigned r0 = pack_bf16x2(mul_nopack(__uint_as_float(w0[e2] << 16), mul), mul_nopack(__uint_as_float(w0[e2] & 0xFFFF0000u), mul));
                        const unsigned r1 = pack_bf16x2(mul_nopack(__uint_as_float(w1[e2] << 16), mul), mul_nopack(__uint_as_float(w1[e2] & 0xFFFF0000u), mul));
                        const bool b0 = ((r0 ^ o0[e2]) & 0x7FFF7FFFu) != 0, b1 = ((r1 ^ o1[e2]) & 0x7FFF7FFFu) != 0;
                        if (b0 || b1) { ++bad; if ((tid & 63) >= 48) ++bad_hi_lanes; }
                    }
                }
            }
        } else if (mode == 1) {  // MFMA chain
            f32x16_t acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            bf16x8_t a, b; for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(tid + i); b[i] = (__bf16)(float)(tid - i); }
            for (int rep = 0; rep < reps * 6; ++rep) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            for (int i = 0; i < 16; ++i) sink += acc[i];
        } else if (mode == 2) {  // LDS reads
            u32x4_t s = {0u, 0u, 0u, 0u};
            for (int rep = 0; rep < reps * 12; ++rep) { const u32x4_t t = *reinterpret_cast<volatile u32x4_t*>(lds + 16384 + ((tid * 16 + rep * 64) & 16383)); s += t; }
            sink += (float)(s[0] + s[1] + s[2] + s[3]);
        } else if (mode == 3) {  // plain VALU
            float x = (float)tid;
            for (int rep = 0; rep < reps * 40; ++rep) x = x * 1.0001f + 0.5f;
            sink += x;
        } else if (mode == 4) {  // transcendental VALU
            float x = (float)tid * 0.001f;
            for (int rep = 0; rep < reps * 10; ++rep) x = __builtin_amdgcn_sinf(x) + 0.3f;
            sink += x;
        } else {  // sleep
            for (int rep = 0; rep < reps; ++rep) __builtin_amdgcn_s_sleep(64);
        }
    }
    if (bad) { atomicAdd(&stats[0], bad); atomicAdd(&stats[1], bad_hi_lanes); atomicAdd(&stats[2], 1u); }
    if (sink == 12345.678f) stats[3] = 1;
}

int main(int argc, char** argv) {
    const int nvec = 1 << 16, reps = argc > 1 ? atoi(argv[1]) : 400, phases = 8;
    std::vector<uint32_t> h((size_t)nvec * 8);
    uint32_t s = 12345u;
    for (auto& w : h) {  // two random bf16 in [-2, 2] per word, never zero / denormal
        uint32_t lohi[2];
        for (int k = 0; k < 2; ++k) { s = s * 1664525u + 1013904223u; const uint32_t mant = (s >> 9) & 0x7F, ex = 120 + ((s >> 20) % 8), sg = (s >> 31); lohi[k] = (sg << 15) | (ex << 7) | mant; }
        w = lohi[0] | (lohi[1] << 16);
    }
    uint32_t* d_in; unsigned* d_stats;
    hipMalloc(&d_in, h.size() * 4); hipMalloc(&d_stats, 16);
    hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    const char* names[] = {"sleep", "mfma", "lds", "valu", "trans"};
    auto run = [&](int asm_victim, int grid, int mode) {
        hipMemset(d_stats, 0, 16);
        probe<<<grid, 256, 56 * 1024>>>(d_in, nvec, d_stats, mode, reps, -1, 0.1803368801f, 1.f, phases, asm_victim);
        hipDeviceSynchronize();
        unsigned st[4]; hipMemcpy(st, d_stats, 16, hipMemcpyDeviceToHost);
        static const char* vn[] = {"c++ (compiler-generated)", "asm1 as generated", "asm2 nop between pk", "asm3 nop before use", "asm4 pk2 other dst", "asm5 only pk1",
                                   "asm6 only pk2 (no op_sel)", "asm7 nop3 before use", "asm8 pk1 without op_sel", "asm9 pk_fma with op_sel"};
        printf("%-26s grid %4d neighbour %-5s: wrong words %5u (in lanes >= 48: %5u) in %4u threads  [%s]\n", vn[asm_victim], grid, names[mode], st[0], st[1], st[2],
               hipGetErrorString(hipGetLastError()));
        fflush(stdout);
    };
    printf("## 1. which neighbour: the extracted sequence and the compiler-generated staging code, grids of one fill (512) and of four (2048)\n");
    for (int asm_victim : {1, 0})
        for (int grid : {512, 2048})
            for (int mode = 0; mode < 5; ++mode) run(asm_victim, grid, mode);
    printf("## 2. which instruction (neighbour = MFMA, grid 2048, 3 trials each)\n");
    for (int asm_victim = 1; asm_victim < 10; ++asm_victim)
        for (int trial = 0; trial < 3; ++trial) run(asm_victim, 2048, 1);
    return 0;
}
