// Probe: what does ONE dependent launch cost on this chip, beyond the work of its workgroups?
// A hipGraph of N kernel nodes in a chain (each depends on the one before, like the pipeline's launches), replayed; time per node for
//   - an empty kernel (1 workgroup; 256 workgroups of 256 threads; 256 workgroups of 1024 threads),
//   - the same grid with the LDS of the fused FFN kernel (131 KB: one workgroup per CU),
//   - a grid whose workgroups each WRITE `wkb` KB (the end-of-kernel write-back of dirty L2 lines is inside the launch),
//   - read-only nodes, and nodes of one load + one store per thread whose source is static or the previous node's output (the producer /
//     consumer pattern of K4-split and the fold kernels: reading what the node before wrote costs nothing extra).
// DESIGN.md section 9.0 prices the pipeline's launch boundaries with these numbers.   build: make probe-launch   run: build/launch_floor_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_empty(int* p) { if (p && threadIdx.x == 4096) p[0] = 1; }

__global__ __launch_bounds__(256, 1) void k_fat(int* p, int n) {  // ~500 VGPRs, dynamic LDS
    extern __shared__ int sm[];
    float a[440];
#pragma unroll
    for (int i = 0; i < 440; ++i) a[i] = (float)(threadIdx.x + i);
    if (n == 12345) {  // never: keeps the registers and the LDS alive
#pragma unroll
        for (int i = 0; i < 440; ++i) sm[(threadIdx.x + i) & 1023] += (int)a[i];
        p[threadIdx.x] = sm[threadIdx.x];
    }
}

__global__ void k_write(uint4* dst, int vec_per_wg) {
    uint4* d = dst + (size_t)blockIdx.x * vec_per_wg;
    for (int i = threadIdx.x; i < vec_per_wg; i += blockDim.x) d[i] = make_uint4(i, 1, 2, 3);
}
// reads only (the sum is stored by nobody: the compare never matches)
__global__ void k_read(const uint4* src, unsigned* sink, int vec_per_wg) {
    const uint4* s = src + (size_t)blockIdx.x * vec_per_wg;
    unsigned acc = 0;
    for (int i = threadIdx.x; i < vec_per_wg; i += blockDim.x) { const uint4 v = s[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) sink[threadIdx.x] = acc;
}
// one load, one store per thread, nothing else
__global__ void k_copy1(const uint4* src, uint4* dst) { dst[blockIdx.x * blockDim.x + threadIdx.x] = src[blockIdx.x * blockDim.x + threadIdx.x]; }

template <typename F>
static double chain(const char* label, int nodes, F enqueue) {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < nodes; ++i) enqueue(s, i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    printf("%-78s %7.2f us per node (%d nodes)\n", label, best * 1e3 / nodes, nodes);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(s));
    return best * 1e3 / nodes;
}

int main() {
    const int N = 400;
    int* dp;
    CK(hipMalloc(&dp, 1 << 20));
    uint4 *b0, *b1;
    const size_t bufbytes = (size_t)256 * 512 * 1024;  // up to 512 KB per workgroup
    CK(hipMalloc(&b0, bufbytes)); CK(hipMalloc(&b1, bufbytes));
    CK(hipMemset(b0, 0, bufbytes)); CK(hipMemset(b1, 0, bufbytes));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fat), hipFuncAttributeMaxDynamicSharedMemorySize, 131 * 1024));
    chain("empty kernel, 1 workgroup x 64", N, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, dp); });
    chain("empty kernel, 256 workgroups x 256", N, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, dp); });
    chain("empty kernel, 256 workgroups x 1024", N, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(1024), 0, s, dp); });
    chain("empty kernel, 2048 workgroups x 256", N, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_empty, dim3(2048), dim3(256), 0, s, dp); });
    chain("empty kernel with 131 KB of LDS per workgroup, 256 workgroups x 256", N, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_fat, dim3(256), dim3(256), 131 * 1024, s, dp, 0); });
    uint4* stat;
    CK(hipMalloc(&stat, bufbytes)); CK(hipMemset(stat, 0, bufbytes));
    chain("256 x 1024 READ-ONLY 16 KB each of a buffer nobody writes", N, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_read, dim3(256), dim3(1024), 0, s, stat, (unsigned*)dp, 1024); });
    chain("256 x 1024 READ-ONLY 96 KB each of a buffer nobody writes", N, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_read, dim3(256), dim3(1024), 0, s, stat, (unsigned*)dp, 6144); });
    chain("256 x 1024 one load + one store per thread, static source, alternating destinations", N, [&](hipStream_t s, int i) { hipLaunchKernelGGL(k_copy1, dim3(256), dim3(1024), 0, s, stat, (i & 1) ? b1 : b0); });
    chain("256 x 1024 one load + one store per thread, previous node's output", N, [&](hipStream_t s, int i) { hipLaunchKernelGGL(k_copy1, dim3(256), dim3(1024), 0, s, (i & 1) ? b0 : b1, (i & 1) ? b1 : b0); });
    chain("1 x 64 one load + one store per thread, previous node's output", N, [&](hipStream_t s, int i) { hipLaunchKernelGGL(k_copy1, dim3(1), dim3(64), 0, s, (i & 1) ? b0 : b1, (i & 1) ? b1 : b0); });
    for (int wkb : {16, 96, 256}) {
        char lab[160];
        const int vec = wkb * 1024 / 16;
        snprintf(lab, sizeof lab, "256 x 256 each WRITE %d KB (%.1f MB per node)", wkb, 256.0 * wkb / 1024);
        chain(lab, N, [&](hipStream_t s, int i) { hipLaunchKernelGGL(k_write, dim3(256), dim3(256), 0, s, (i & 1) ? b1 : b0, vec); });
    }
    return 0;
}
