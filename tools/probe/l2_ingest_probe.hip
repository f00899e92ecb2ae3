// Probe: how many bytes per clock does ONE compute unit receive from L2 when every CU of the chip streams at once?
// The fused FFN kernels (kernels_ffn*.hip) and the head-split cross-attention stream their weights — a few MB that every workgroup reads, so they are
// L2 hits after the first touch — through exactly this path; DESIGN.md section 9.1 prices those kernels against the rate this probe prints.
//
// Every workgroup reads the same `span` bytes `reps` times (16-byte loads, 4 KiB per workgroup per instruction, `UNROLL` loads in flight per lane), either
// into registers (mode 0) or straight into LDS by LDS-DMA (mode 1: buffer_load ... lds, the ring the FFN kernels use).  Own-region mode (2) gives every
// workgroup its own span instead: no sharing, the HBM / MALL side.
//   build: make probe-ingest     run (GPU box): build/l2_ingest_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int UNROLL = 8;

template <int MODE>
__global__ __launch_bounds__(256, 1) void ingest(const uint4* __restrict__ src, size_t span_vec /*uint4 per span*/, int reps, size_t wg_stride_vec,
                                                 unsigned* __restrict__ sink, unsigned long long* __restrict__ cycles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const uint4* base = src + (size_t)blockIdx.x * wg_stride_vec;
    uint4 acc = make_uint4(0, 0, 0, 0);
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (MODE == 1) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(base), 0, (unsigned)(span_vec * 16), 0x00020000);
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        for (int r = 0; r < reps; ++r)
            for (size_t i = 0; i < span_vec; i += 256 * UNROLL) {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u)  // each wave's 64 lanes x 16 B = 1 KiB per instruction, into its own 8 KiB of LDS
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(smem + wave * (UNROLL * 1024) + u * 1024), 16, (unsigned)((i + u * 256 + wave * 64 + lane) * 16), 0, 0, 0);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UNROLL) : "memory");  // one batch stays in flight behind the one being waited for
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc.x = reinterpret_cast<unsigned*>(smem)[tid];
    } else {
        for (int r = 0; r < reps; ++r)
            for (size_t i = tid; i < span_vec; i += 256 * UNROLL) {
                uint4 v[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) v[u] = base[i + u * 256];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[blockIdx.x * 256 + tid] = acc.x;  // keeps the loads alive
}

template <int MODE>
static void run(const char* label, const uint4* d, size_t span, int reps, int wgs, bool own, unsigned* sink, unsigned long long* dcyc) {
    const size_t span_vec = span / 16;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = MODE == 1 ? 4 * UNROLL * 1024 : 0;
    for (int warm = 0; warm < 2; ++warm) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(ingest<MODE>, dim3(wgs), dim3(256), lds, 0, d, span_vec, reps, own ? span_vec : (size_t)0, sink, dcyc);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
    }
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> cyc(wgs);
    CK(hipMemcpy(cyc.data(), dcyc, sizeof(unsigned long long) * wgs, hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto c : cyc) mean += (double)c;
    mean /= wgs;
    const double bytes = (double)span * reps * wgs;
    // the cycle counter of s_memtime / readcyclecounter ticks at 100 MHz on gfx9: report the event time and derive bytes per shader clock from a nominal 2.1 GHz
    printf("%-34s wgs %5d span %7.2f MB x %3d: %8.1f us  %6.2f TB/s aggregate  %5.1f B/clk per CU at 2.1 GHz (%d CUs busy)\n", label, wgs, span / 1048576.0, reps, ms * 1e3,
           bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 2.1e9 / (wgs < 256 ? wgs : 256), wgs < 256 ? wgs : 256);
    (void)mean;
}

int main() {
    const size_t maxspan = (size_t)8 << 20;
    const int maxwg = 1024;
    uint4* d;
    unsigned* sink;
    unsigned long long* dcyc;
    CK(hipMalloc(&d, maxspan * 64));  // own-region mode: up to 64 spans of 8 MB
    CK(hipMemset(d, 1, maxspan * 64));
    CK(hipMalloc(&sink, sizeof(unsigned) * maxwg * 256));
    CK(hipMalloc(&dcyc, sizeof(unsigned long long) * maxwg));
    const size_t spans[] = {(size_t)590 << 10, (size_t)2416 << 10, (size_t)4288 << 10};  // K4-split's share of a block's weights, its whole block, the vocoder's block
    for (size_t span : spans) {
        const size_t sp = span / (256 * UNROLL * 16) * (256 * UNROLL * 16);
        const int reps = (int)(((size_t)48 << 20) / sp) + 1;
        for (int wgs : {1, 32, 256, 512}) {
            run<0>("shared span, loads to registers", d, sp, reps, wgs, false, sink, dcyc);
        }
        run<1>("shared span, LDS-DMA", d, sp, reps, 256, false, sink, dcyc);
    }
    run<0>("own span per workgroup (no sharing)", d, (size_t)2 << 20, 4, 256, true, sink, dcyc);
    run<1>("own span per workgroup, LDS-DMA", d, (size_t)2 << 20, 4, 256, true, sink, dcyc);
    return 0;
}
