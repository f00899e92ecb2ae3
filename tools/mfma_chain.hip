// Micro-measurement (gfx950): issue cost of v_mfma_f32_32x32x16_bf16 in a chain of dependent accumulations against independent accumulators,
// one wave per SIMD (256 threads per workgroup, one workgroup per CU), the occupancy K4 runs at.  Build: hipcc --offload-arch=gfx950 -O3
// tools/mfma_chain.hip -o tools/_bin/mfma_chain.  Prints shader cycles per MFMA for each pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int NACC>
__global__ __launch_bounds__(256) void chain(float* out, long long* cyc, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x * 3 + i)); }
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 24; ++j) acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j % NACC], 0, 0, 0);
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
static void run(const char* name, float* out, long long* cyc, int wgs) {
    const int iters = 200;
    chain<NACC><<<wgs, 256>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    chain<NACC><<<wgs, 256>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[1024]; hipMemcpy(h, cyc, sizeof(long long) * wgs, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < wgs; ++i) avg += (double)h[i]; avg /= wgs;
    const double n = 24.0 * iters;
    printf("%-28s wgs=%4d: %.1f clock64 ticks per MFMA, %.2f ns per MFMA (launch %.1f us) -> ticks at %.0f MHz\n", name, wgs, avg / n, ms * 1e6 / n, ms * 1e3,
           avg / (ms * 1e3));
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    for (int wgs : {1, 256}) {
        run<1>("1 accumulator (dependent)", out, cyc, wgs);
        run<2>("2 accumulators", out, cyc, wgs);
        run<3>("3 accumulators", out, cyc, wgs);
        run<4>("4 accumulators", out, cyc, wgs);
        run<12>("12 accumulators", out, cyc, wgs);
    }
    return 0;
}
