// Probe: cycles per v_mfma_f32_32x32x16_bf16 at one wave per SIMD for (a) one dependent accumulation chain in arch VGPRs,
// (b) the same in AGPRs, (c) two alternating chains, (d) four, (e) one chain with a ds_read_b128 per MFMA, (f) one chain in blocks of
// four separated by s_waitcnt lgkmcnt(0) + s_nop 7 (the block shape of kernels_ffn.hip).   build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u + i;
    __syncthreads();
    bf16x8 a, b, c2 = {};
    float fx = 1.0001f;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (lane + j)); b[j] = (__bf16)(0.002f * (lane - j)); }
    f32x16 acc0, acc1, acc2, acc3;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; acc2[i] = 0.f; acc3[i] = 0.f; }
    const unsigned ad = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + lane * 16;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) asm volatile(REP64("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t") : "+v"(acc0) : "v"(a), "v"(b));
        if (MODE == 1) asm volatile(REP64("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t") : "+a"(acc0) : "v"(a), "v"(b));
        if (MODE == 2) asm volatile(REP16(REP4("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t")) : "+a"(acc0), "+a"(acc1) : "v"(a), "v"(b));
        if (MODE == 3) asm volatile(REP16(REP4("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\tv_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %4, %5, %3\n\t")) : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3) : "v"(a), "v"(b));
        if (MODE == 4) asm volatile(REP64("ds_read_b128 %1, %4\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t") "s_waitcnt lgkmcnt(0)" : "+v"(acc0), "=&v"(c2) : "v"(a), "v"(b), "v"(ad));
        if (MODE == 5) asm volatile(REP16(REP4("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t") "s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 2\n\t") : "+v"(acc0) : "v"(a), "v"(b));
        if (MODE == 6) asm volatile(REP16(REP4("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t") "s_waitcnt lgkmcnt(0)\n\t") : "+v"(acc0) : "v"(a), "v"(b));
        if (MODE == 7) asm volatile(REP16("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mul_f32 %1, %1, %1\n\tv_mul_f32 %1, %1, %1\n\tv_mul_f32 %1, %1, %1\n\tv_mul_f32 %1, %1, %1\n\t" "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t") : "+v"(acc0), "+v"(fx) : "v"(a), "v"(b));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)c2[0] + fx;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name, int nmfma_per_iter) {
    float* out; unsigned long long* cyc;
    const int nb = 256, iters = 200;
    hipMalloc(&out, nb * 256 * 4); hipMalloc(&cyc, nb * 8);
    probe<MODE><<<nb, 256>>>(out, cyc, iters);
    probe<MODE><<<nb, 256>>>(out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nb);
    hipMemcpy(h.data(), cyc, nb * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-60s median %.1f cycles per MFMA\n", name, (double)h[nb / 2] / iters / nmfma_per_iter);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0>("one chain, accumulator in arch VGPRs", 64);
    run<1>("one chain, accumulator in AGPRs", 64);
    run<2>("two alternating chains (AGPR)", 128);
    run<3>("four alternating chains (AGPR)", 256);
    run<4>("one chain (VGPR) + one ds_read_b128 per MFMA", 64);
    run<5>("one chain (VGPR), blocks of 4 + lgkmcnt(0) + s_nop 7,2", 64);
    run<6>("one chain (VGPR), blocks of 4 + lgkmcnt(0)", 64);
    run<7>("one chain (VGPR), 4 VALU behind every 4th MFMA", 64);
    return 0;
}
