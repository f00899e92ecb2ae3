// Probe: four 32 KiB stages into a 128 KiB LDS ring by LDS-DMA exactly as ffn_issue does (4 waves x 8 KiB pieces each),
// then the ring copied out.  Prints every KiB piece whose content is not the source's.   build: make probe2; run on a GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define STN_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (unsigned)bytes, 0x00020000);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned char* lds_dst, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, STN_LDS_PTR(lds_dst), 16, voff, soff, 0, 0);
}
template <int SB, int PER>
__global__ __launch_bounds__(256, 1) void probe(const unsigned* src, unsigned* out, int extra) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < (4 * SB + extra) / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0xDEADBEEFu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(src, (size_t)4 * SB);
    const unsigned voff = (unsigned)(wave * PER * 1024 + lane * 16);
    for (int sg = 0; sg < 4; ++sg) {
        unsigned char* dst = smem + (sg & 3) * SB + wave * (PER * 1024);
#pragma unroll
        for (int j = 0; j < PER; ++j) dma16(rs, dst + j * 1024, voff + j * 1024, sg * SB);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 4 * SB / 4; i += 256) out[i] = reinterpret_cast<unsigned*>(smem)[i];
}
int main() {
    constexpr int SB = 32768, PER = 8;
    const size_t n = (size_t)4 * SB / 4;
    std::vector<unsigned> h(n), o(n);
    for (size_t i = 0; i < n; ++i) h[i] = (unsigned)i * 2654435761u + 12345u;
    unsigned *d, *dout;
    hipMalloc(&d, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int extra : {0, 6144, 12288, 32768}) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<SB, PER>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipMemset(dout, 0, n * 4);
        probe<SB, PER><<<dim3(8), dim3(256), 4 * SB + extra>>>(d, dout, extra);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int p = 0; p < 4 * SB / 1024; ++p) {
            int nb = 0; unsigned first = 0;
            for (int i = 0; i < 256; ++i) if (o[p * 256 + i] != h[p * 256 + i]) { if (!nb) first = o[p * 256 + i]; ++nb; }
            if (nb) { ++bad; printf("  extra %d: piece %d (stage %d, piece %d, LDS 0x%x): %d/256 words wrong, first = 0x%08x\n", extra, p, p / 32, p % 32, p * 1024, nb, first); }
        }
        printf("extra LDS %d: %s, %d bad pieces of %d\n", extra, hipGetErrorString(e), bad, 4 * SB / 1024);
    }
    return 0;
}
