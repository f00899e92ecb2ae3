// Micro-measurement (gfx950): issue cost of v_mfma_f32_32x32x16_bf16, one wave per SIMD (256 threads per workgroup, one workgroup per CU — the
// occupancy K4 runs at): chains of dependent accumulations against independent accumulators, accumulator in the arch-VGPR or the AGPR half of the
// file, operands constant or different for every MFMA.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_chain.hip -o tools/_bin/mfma_chain.
// Prints shader cycles per MFMA (clock64) and the clock the chip held (cycles / wall time of the launch).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

__device__ __forceinline__ bf16x8 mk(int seed) {
    bf16x8 v;
    for (int i = 0; i < 8; ++i) v[i] = (__bf16)(0.001f * (float)((threadIdx.x * 7 + seed * 13 + i * 3) % 97) - 0.05f);
    return v;
}

// MODE 0: NACC accumulators (compiler-placed: AGPRs), constant operands
// MODE 1: one chained accumulator in arch VGPRs ("+v"), 24 different B operands (K4 phase 1)
// MODE 2: one chained accumulator in AGPRs ("+a"), 24 different B operands
// MODE 3: 12 AGPR accumulators, A operand different for every MFMA out of 8, B one of two (K4 phase 2)
// MODE 4: as 1, but A also different for every MFMA (8 fragment registers in rotation, as the ring's reads deliver them)
template <int MODE, int NACC>
__global__ __launch_bounds__(256) void chain(float* out, long long* cyc, int iters) {
    bf16x8 x[24], f[8];
    for (int i = 0; i < 24; ++i) x[i] = mk(i);
    for (int i = 0; i < 8; ++i) f[i] = mk(100 + i);
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 24; ++j) acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], x[0], acc[j % NACC], 0, 0, 0);
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 24; ++j) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(f[0]), "v"(x[j]));
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 24; ++j) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[0]) : "v"(f[0]), "v"(x[j]));
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 24; ++j) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[(j >> 2) * 2 + (j & 1)]) : "v"(f[j & 7]), "v"(x[(j >> 1) & 1]));
        } else if constexpr (MODE == 4) {
#pragma unroll
            for (int j = 0; j < 24; ++j) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(f[j & 7]), "v"(x[j]));
        } else if constexpr (MODE == 5) {  // K4's block: four chained MFMAs with the M0 save (s_mov from m0) behind the first
#pragma unroll
            for (int j = 0; j < 24; j += 4) {
                unsigned keep;
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\ts_mov_b32 %1, m0\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %4, %0\n\t"
                             "v_mfma_f32_32x32x16_bf16 %0, %2, %5, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %6, %0\n\ts_waitcnt lgkmcnt(0)"
                             : "+v"(acc[0]), "=&s"(keep) : "v"(f[0]), "v"(x[j]), "v"(x[j + 1]), "v"(x[j + 2]), "v"(x[j + 3]) : "memory");
            }
        } else if constexpr (MODE == 6) {  // ... with M0 saved, written and restored (no DMA between)
#pragma unroll
            for (int j = 0; j < 24; j += 4) {
                unsigned keep;
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\ts_mov_b32 %1, m0\n\ts_mov_b32 m0, %7\n\ts_nop 0\n\ts_mov_b32 m0, %1\n\t"
                             "v_mfma_f32_32x32x16_bf16 %0, %2, %4, %0\n\t"
                             "v_mfma_f32_32x32x16_bf16 %0, %2, %5, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %6, %0\n\ts_waitcnt lgkmcnt(0)"
                             : "+v"(acc[0]), "=&s"(keep) : "v"(f[0]), "v"(x[j]), "v"(x[j + 1]), "v"(x[j + 2]), "v"(x[j + 3]), "s"(iters + j) : "memory");
            }
        } else if constexpr (MODE == 7) {  // the block without touching M0
#pragma unroll
            for (int j = 0; j < 24; j += 4) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %3, %0\n\t"
                             "v_mfma_f32_32x32x16_bf16 %0, %1, %4, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %5, %0\n\ts_waitcnt lgkmcnt(0)"
                             : "+v"(acc[0]) : "v"(f[0]), "v"(x[j]), "v"(x[j + 1]), "v"(x[j + 2]), "v"(x[j + 3]) : "memory");
            }
        } else if constexpr (MODE == 9 || MODE == 10 || MODE == 11) {  // K4's phase-2 stage as the MFMA-only timing variant issues it
            float ha = (float)it, hb = 1.f - (float)it;
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                unsigned keep, gw; float ta, tb;
                if constexpr (MODE == 9)
                    asm volatile("v_mfma_f32_32x32x16_bf16 %[ya], %[f0], %[g0], %[ya]\n\ts_mov_b32 %[keep], m0\n\tv_mfma_f32_32x32x16_bf16 %[yb], %[f2], %[g0], %[yb]\n\t"
                                 "v_mov_b32 %[ta], %[ha]\n\tv_mov_b32 %[tb], %[hb]\n\tv_mfma_f32_32x32x16_bf16 %[ya], %[f1], %[g1], %[ya]\n\t"
                                 "v_mfma_f32_32x32x16_bf16 %[yb], %[f3], %[g1], %[yb]\n\tv_cvt_pk_bf16_f32 %[gw], %[ta], %[tb]\n\ts_nop 7\n\ts_waitcnt lgkmcnt(0)"
                                 : [ya] "+a"(acc[2 * b]), [yb] "+a"(acc[2 * b + 1]), [keep] "=&s"(keep), [ta] "=&v"(ta), [tb] "=&v"(tb), [gw] "=&v"(gw)
                                 : [f0] "v"(f[(4 * b) & 7]), [f1] "v"(f[(4 * b + 1) & 7]), [f2] "v"(f[(4 * b + 2) & 7]), [f3] "v"(f[(4 * b + 3) & 7]), [g0] "v"(x[0]), [g1] "v"(x[1]),
                                   [ha] "v"(ha), [hb] "v"(hb) : "memory");
                else if constexpr (MODE == 10)  // ... without the s_nop 7
                    asm volatile("v_mfma_f32_32x32x16_bf16 %[ya], %[f0], %[g0], %[ya]\n\ts_mov_b32 %[keep], m0\n\tv_mfma_f32_32x32x16_bf16 %[yb], %[f2], %[g0], %[yb]\n\t"
                                 "v_mov_b32 %[ta], %[ha]\n\tv_mov_b32 %[tb], %[hb]\n\tv_mfma_f32_32x32x16_bf16 %[ya], %[f1], %[g1], %[ya]\n\t"
                                 "v_mfma_f32_32x32x16_bf16 %[yb], %[f3], %[g1], %[yb]\n\tv_cvt_pk_bf16_f32 %[gw], %[ta], %[tb]\n\ts_waitcnt lgkmcnt(0)"
                                 : [ya] "+a"(acc[2 * b]), [yb] "+a"(acc[2 * b + 1]), [keep] "=&s"(keep), [ta] "=&v"(ta), [tb] "=&v"(tb), [gw] "=&v"(gw)
                                 : [f0] "v"(f[(4 * b) & 7]), [f1] "v"(f[(4 * b + 1) & 7]), [f2] "v"(f[(4 * b + 2) & 7]), [f3] "v"(f[(4 * b + 3) & 7]), [g0] "v"(x[0]), [g1] "v"(x[1]),
                                   [ha] "v"(ha), [hb] "v"(hb) : "memory");
                else  // ... MFMAs only, same accumulator order
                    asm volatile("v_mfma_f32_32x32x16_bf16 %[ya], %[f0], %[g0], %[ya]\n\tv_mfma_f32_32x32x16_bf16 %[yb], %[f2], %[g0], %[yb]\n\t"
                                 "v_mfma_f32_32x32x16_bf16 %[ya], %[f1], %[g1], %[ya]\n\t"
                                 "v_mfma_f32_32x32x16_bf16 %[yb], %[f3], %[g1], %[yb]\n\ts_waitcnt lgkmcnt(0)"
                                 : [ya] "+a"(acc[2 * b]), [yb] "+a"(acc[2 * b + 1])
                                 : [f0] "v"(f[(4 * b) & 7]), [f1] "v"(f[(4 * b + 1) & 7]), [f2] "v"(f[(4 * b + 2) & 7]), [f3] "v"(f[(4 * b + 3) & 7]), [g0] "v"(x[0]), [g1] "v"(x[1])
                                 : "memory");
            }
        } else {  // MODE 8: the block with M0 written (clobber declared), not saved
#pragma unroll
            for (int j = 0; j < 24; j += 4) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %3, %0\n\t"
                             "v_mfma_f32_32x32x16_bf16 %0, %1, %4, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %5, %0\n\ts_waitcnt lgkmcnt(0)"
                             : "+v"(acc[0]) : "v"(f[0]), "v"(x[j]), "v"(x[j + 1]), "v"(x[j + 2]), "v"(x[j + 3]), "s"(iters + j) : "memory", "m0");
            }
        }
    }
    const long long t1 = clock64();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; cyc[1024 + blockIdx.x] = (long long)(r1 - r0); }
}

template <int MODE, int NACC>
static void run(const char* name, float* out, long long* cyc, int wgs) {
    const int iters = 200;
    chain<MODE, NACC><<<wgs, 256>>>(out, cyc, iters);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    chain<MODE, NACC><<<wgs, 256>>>(out, cyc, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2048]; (void)hipMemcpy(h, cyc, sizeof(long long) * 2048, hipMemcpyDeviceToHost);
    double avg = 0, rt = 0; for (int i = 0; i < wgs; ++i) { avg += (double)h[i]; rt += (double)h[1024 + i]; } avg /= wgs; rt /= wgs;
    const double n = 24.0 * iters;
    // in-kernel clock = s_memtime ticks / s_memrealtime ticks x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6)
    printf("%-64s wgs=%4d: %5.1f cycles per MFMA, %6.2f ns per MFMA in the loop (%.1f us; launch %.1f us), in-kernel clock %.0f MHz\n", name, wgs, avg / n, rt * 10.0 / n,
           rt / 100.0, ms * 1e3, avg / rt * 100.0);
}

int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&cyc, 2048 * 8);
    for (int wgs : {1, 256}) {
        run<0, 1>("1 accumulator (dependent chain), constant operands", out, cyc, wgs);
        run<0, 2>("2 accumulators, constant operands", out, cyc, wgs);
        run<0, 12>("12 accumulators, constant operands", out, cyc, wgs);
        run<1, 1>("chain in arch VGPRs, 24 different B operands (K4 phase 1)", out, cyc, wgs);
        run<2, 1>("chain in AGPRs, 24 different B operands", out, cyc, wgs);
        run<4, 1>("chain in arch VGPRs, A and B different for every MFMA", out, cyc, wgs);
        run<3, 12>("12 AGPR accumulators, 8 A operands in rotation (K4 phase 2)", out, cyc, wgs);
        run<7, 1>("K4 block (4 chained MFMAs + s_waitcnt), M0 untouched", out, cyc, wgs);
        run<5, 1>("K4 block + s_mov sN, m0 (the M0 save)", out, cyc, wgs);
        run<6, 1>("K4 block + M0 saved, written, restored", out, cyc, wgs);
        run<8, 1>("K4 block + M0 written, declared as a clobber", out, cyc, wgs);
        run<11, 12>("K4 phase-2 stage, MFMAs only (ya, yb, ya, yb per block)", out, cyc, wgs);
        run<10, 12>("K4 phase-2 stage as timing variant 8 issues it, no s_nop 7", out, cyc, wgs);
        run<9, 12>("K4 phase-2 stage as timing variant 8 issues it", out, cyc, wgs);
    }
    return 0;
}
