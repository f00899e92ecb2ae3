// kernels_ffn.hip — K4: the pointwise pair of a ConvNeXt block in ONE launch, gfx950.
//
//   x[m][:] <- (x[m][:] + gamma * (W2 . GELU(W1 . xn[m] + b1) + b2) + rowvec[seq(m)]) * keep(m)
//
// xn = LayerNorm(dwconv(x)) in the 16-bit activation format, x the fp32 residual stream.  The 4C-wide hidden activation
// never exists in memory: it lives for one 32-unit tile in the accumulator registers of the wave that owns the rows.
// Stands in for the body of vocoder_ort_->Run / vector_est_ort_->Run (/root/reference/cpp/helper.cpp:643-647, 668-672).
//
// Shape of the computation (both GEMMs TRANSPOSED, so that a lane is a ROW of the activation):
//   phase 1   Ht[h][r]  = sum_c W1[h][c] . xn[r][c]      A = W1 tile (32 hidden x 16 c per MFMA), B = xn fragments, held in
//                                                          registers for the whole kernel (C/16 x 4 VGPRs)
//   GELU on the accumulator, rounded to bf16: the result IS the B operand of phase 2 (CDNA4 "accumulator as the next
//             operand": registers 8s..8s+7 of lane half hf are hidden units 16s + 8(j>>2) + 4hf + (j&3), so W2 is stored
//             with its k order permuted to match — launch_repack_frag_acc)
//   phase 2   Yt[n][r] += sum_h W2[n][h] . G[r][h]        A = W2 tile (32 out channels x 16 hidden), all C/32 tiles of Yt stay
//                                                          in the accumulator file (C/2 AGPRs per lane)
//   epilogue  a lane holds 4 CONSECUTIVE channels of ITS row per register quad: residual update with 16-byte accesses
//             straight from the accumulators, no LDS transpose.
// A workgroup = 4 waves x 32 rows = 128 rows, one wave per SIMD with the whole 512-register file.  Both weight matrices are
// pre-packed in MFMA fragment order (one stage = one hidden tile of one matrix = C x 64 bytes, a run of contiguous KiBs), so
// every LDS-DMA wave-instruction moves one fully contiguous KiB (eight whole cache lines) and every fragment read is a
// lane-linear ds_read_b128 at base + immediate.  The stages stream through a 4-deep ring shared by the four waves, in the
// order W1(0), W1(1), W2(0), W1(2), W2(1), ... so that the GELU of tile t runs in the shadow of phase 1 of tile t+1.
//
// One wave per SIMD means nothing but the wave's own instruction order hides an LDS read or a VALU instruction, and hipcc
// (ROCm 7.2) serialises this loop (one ds_read_b128 -> s_waitcnt -> v_mfma at a time, the GELU behind the MFMAs, a
// vmcnt(0) in front of every LDS read that follows an LDS-DMA, sched_group_barrier ignored).  The stage bodies are therefore
// written as inline-asm BLOCKS of four MFMAs: each block issues the four fragment reads of the NEXT block first, then its
// MFMAs with two (or four) GELU elements in their shadows, and ends with the s_waitcnt lgkmcnt(0) that makes its outputs
// valid (cdna_hip_programming.md 5.7: the loads and their wait in ONE statement, outputs early-clobber).  The ring
// hand-over of stage s+1 (counted vmcnt, raw s_barrier, refill of the stage vacated two stages ago) sits in the MIDDLE of
// stage s, so the last block of a stage can already read the first fragments of the next one: no bubble at a stage seam.
#include "kernels.hpp"
#include "dev_env.hpp"
#include "kernels_dev.hpp"

#include <stdio.h>

namespace stn {

typedef __attribute__((ext_vector_type(4))) float f32x4;

// the product kernel, bf16 operands ...
#define STN_V_NODMA 0
#define STN_V_NOGELU 0
#define STN_V_EARLYRD 0
#ifndef STN_V_ACCA
#define STN_V_ACCA 0
#endif
#define STN_V_F16 0
#include "kernels_ffn_body.inc"
#undef STN_V_F16
// ... and IEEE-half operands (the engine's f16 mode).  Same instruction stream; the GELU inside the blocks is the exp2 form in both
// (|err| <= 5e-4 absolute: about one half-precision ulp at |y| ~ 1, where the two-launch f16 path uses the erf form).
namespace f16k {
#define STN_V_F16 1
#include "kernels_ffn_body.inc"
#undef STN_V_F16
}
#define STN_V_F16 0
#undef STN_V_NODMA
#undef STN_V_NOGELU
#undef STN_V_EARLYRD
// timing-only variants (wrong results; each drops one ingredient of a block to see what it costs — DESIGN.md section 5d).
// Not part of the product build: `make EXTRA=-DSTN_FFN_VARIANTS` compiles them in and STN_FFN_VAR=<1..5> selects one.
#ifdef STN_FFN_VARIANTS
namespace v_nodma {
#define STN_V_NODMA 1
#define STN_V_NOGELU 0
#define STN_V_EARLYRD 0
#ifndef STN_V_ACCA
#define STN_V_ACCA 0
#endif
#include "kernels_ffn_body.inc"
#undef STN_V_NODMA
#undef STN_V_NOGELU
#undef STN_V_EARLYRD
}
namespace v_nogelu {
#define STN_V_NODMA 0
#define STN_V_NOGELU 1
#define STN_V_EARLYRD 0
#ifndef STN_V_ACCA
#define STN_V_ACCA 0
#endif
#include "kernels_ffn_body.inc"
#undef STN_V_NODMA
#undef STN_V_NOGELU
#undef STN_V_EARLYRD
}
namespace v_earlyrd {
#define STN_V_NODMA 0
#define STN_V_NOGELU 0
#define STN_V_EARLYRD 1
#ifndef STN_V_ACCA
#define STN_V_ACCA 0
#endif
#include "kernels_ffn_body.inc"
#undef STN_V_NODMA
#undef STN_V_NOGELU
#undef STN_V_EARLYRD
}
namespace v_nord {
#define STN_V_NODMA 1
#define STN_V_NOGELU 1
#define STN_V_EARLYRD 2
#ifndef STN_V_ACCA
#define STN_V_ACCA 0
#endif
#include "kernels_ffn_body.inc"
#undef STN_V_NODMA
#undef STN_V_NOGELU
#undef STN_V_EARLYRD
}
namespace v_nobar {
#define STN_V_NODMA 0
#define STN_V_NOGELU 0
#define STN_V_EARLYRD 0
#define STN_V_NOBAR 1
#ifndef STN_V_ACCA
#define STN_V_ACCA 0
#endif
#include "kernels_ffn_body.inc"
#undef STN_V_NODMA
#undef STN_V_NOGELU
#undef STN_V_EARLYRD
#undef STN_V_NOBAR
#undef STN_HANDOVER
}
namespace v_nohand {
#define STN_V_NODMA 0
#define STN_V_NOGELU 0
#define STN_V_EARLYRD 0
#define STN_V_NOBAR 2
#ifndef STN_V_ACCA
#define STN_V_ACCA 0
#endif
#include "kernels_ffn_body.inc"
#undef STN_V_NODMA
#undef STN_V_NOGELU
#undef STN_V_EARLYRD
#undef STN_V_NOBAR
#undef STN_HANDOVER
}
namespace v_mfmaonly {
#define STN_V_NODMA 1
#define STN_V_NOGELU 1
#define STN_V_EARLYRD 2
#define STN_V_NOBAR 2
#ifndef STN_V_ACCA
#define STN_V_ACCA 0
#endif
#include "kernels_ffn_body.inc"
#undef STN_V_NODMA
#undef STN_V_NOGELU
#undef STN_V_EARLYRD
#undef STN_V_NOBAR
#undef STN_HANDOVER
}
namespace v_stamp {
#define STN_V_NODMA 0
#define STN_V_NOGELU 0
#define STN_V_EARLYRD 0
#define STN_V_STAGESTAMP 1
#ifndef STN_V_ACCA
#define STN_V_ACCA 0
#endif
#include "kernels_ffn_body.inc"
#undef STN_V_NODMA
#undef STN_V_NOGELU
#undef STN_V_EARLYRD
#undef STN_V_STAGESTAMP
}
namespace v_stamp_mfma {
#define STN_V_NODMA 1
#define STN_V_NOGELU 1
#define STN_V_EARLYRD 2
#define STN_V_NOBAR 2
#define STN_V_STAGESTAMP 1
#ifndef STN_V_ACCA
#define STN_V_ACCA 0
#endif
#include "kernels_ffn_body.inc"
#undef STN_V_NODMA
#undef STN_V_NOGELU
#undef STN_V_EARLYRD
#undef STN_V_NOBAR
#undef STN_V_STAGESTAMP
#undef STN_HANDOVER
}
namespace v_bare {
#define STN_V_NODMA 1
#define STN_V_NOGELU 1
#define STN_V_EARLYRD 0
#ifndef STN_V_ACCA
#define STN_V_ACCA 0
#endif
#include "kernels_ffn_body.inc"
#undef STN_V_NODMA
#undef STN_V_NOGELU
#undef STN_V_EARLYRD
}

#endif

// W2 [N = C][K = I] row-major 16-bit -> phase-2 A fragments in the accumulator-operand k order:
//   piece ((t * C/32 + nt) * 2 + s), lane (r, hf), element j  <-  W2[32nt + r][32t + 16s + 8(j>>2) + 4hf + (j&3)]
__global__ void repack_frag_acc_kernel(const uint16_t* __restrict__ W, int N, int K, uint16_t* __restrict__ Wf) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte element group each
    const int NT2 = N / 32, T = K / 32;
    if (idx >= (int64_t)T * NT2 * 2 * 64) return;
    const int lane = (int)(idx & 63);
    int64_t pc = idx >> 6;
    const int s = (int)(pc & 1); pc >>= 1;
    const int nt = (int)(pc % NT2), t = (int)(pc / NT2);
    const int r = lane & 31, hf = lane >> 5;
    const uint16_t* src = W + (size_t)(32 * nt + r) * K + 32 * t + 16 * s + 4 * hf;
    const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 8);
    reinterpret_cast<uint4*>(Wf)[idx] = make_uint4(lo.x, lo.y, hi.x, hi.y);
}

void launch_repack_frag_acc(hipStream_t s, const void* W, int N, int K, void* Wf) {
    if (N % 32 || K % 32) throw std::invalid_argument("launch_repack_frag_acc: N % 32 and K % 32 must be 0");
    const int64_t n = (int64_t)(K / 32) * (N / 32) * 2 * 64;
    STN_KLAUNCH(repack_frag_acc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, static_cast<const uint16_t*>(W), N, K,
                static_cast<uint16_t*>(Wf));
}

// W1 [N = I][K = C] row-major 16-bit -> phase-1 A fragments: lane (lr, lh) of hidden tile T, k-step ks holds
// W[32T + lr][16ks + 8lh .. +8]; the 64 x 16 bytes of one MFMA operand are one contiguous KiB at ((T * K/16 + ks) * 64 + lane) * 16
__global__ void repack_frag_kernel(const uint16_t* __restrict__ W, int N, int K, uint16_t* __restrict__ Wf) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte piece each
    const int NKS = K / 16;
    if (idx >= (int64_t)(N / 32) * NKS * 64) return;
    const int lane = (int)(idx & 63);
    const int64_t blk = idx >> 6;
    const int ks = (int)(blk % NKS), T = (int)(blk / NKS);
    const int lr = lane & 31, lh = lane >> 5;
    reinterpret_cast<uint4*>(Wf)[idx] = *reinterpret_cast<const uint4*>(W + (size_t)(T * 32 + lr) * K + ks * 16 + lh * 8);
}

void launch_repack_frag(hipStream_t s, const void* W, int N, int K, void* Wf) {
    if (N % 32 || K % 16) throw std::invalid_argument("launch_repack_frag: N % 32 and K % 16 must be 0");
    const int64_t n = (int64_t)(N / 32) * (K / 16) * 64;
    STN_KLAUNCH(repack_frag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, static_cast<const uint16_t*>(W), N, K, static_cast<uint16_t*>(Wf));
}

void launch_ffn_pack(hipStream_t s, const void* W1, const void* W2, int C, int I, void* tmp, void* wseq, int S) {
    if (C % 64 || I % 64 || S < 1 || (I / 32) % (2 * S)) throw std::invalid_argument("launch_ffn_pack: C % 64, I % 64 and (I / 32) % (2 S) must be 0");
    const size_t SB = (size_t)C * 64, T = (size_t)I / 32 / S;  // T: hidden tiles per stream
    unsigned char* t1 = static_cast<unsigned char*>(tmp);
    unsigned char* t2 = t1 + (size_t)I * C * 2;
    launch_repack_frag(s, W1, I, C, t1);       // hidden tile t = bytes [t*SB, (t+1)*SB)
    launch_repack_frag_acc(s, W2, C, I, t2);   // likewise
    unsigned char* out = static_cast<unsigned char*>(wseq);
    for (size_t sp = 0; sp < (size_t)S; ++sp)
        for (size_t v = 0; v < 2 * T; ++v) {
            const bool w2 = v == 2 * T - 1 || (v >= 2 && (v & 1) == 0);
            const size_t tile = sp * T + (v == 0 ? 0 : v == 2 * T - 1 ? T - 1 : w2 ? (v - 2) / 2 : (v + 1) / 2);
            stn_check_hip(hipMemcpyAsync(out + (sp * 2 * T + v) * SB, (w2 ? t2 : t1) + tile * SB, SB, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync(ffn_pack)");
        }
}

// LDS of one workgroup: the 4-stage ring, b2 / gamma / b1 behind it; the split form's epilogue re-uses it as four wave-private
// row images of 32 x (2C + 16) bytes
static size_t ffn_lds_bytes(int C, int I) {
    size_t ring = (size_t)4 * C * 64 + (size_t)(I + 2 * C) * 4;
    if (ffn_lds_has_side(C)) ring = (size_t)ffn_lds_side_offset(C, I) + (size_t)C * 64;  // a fifth buffer: wave 0's rows of the slab in the prologue (kernels.hpp)
    const size_t img = (size_t)4 * 32 * (C * 2 + 16);
    return ring > img ? ring : img;
}

bool ffn_fused_supported(int dtype, int C, int I) {
    return is_half(dtype) && (C == 384 || C == 512) && I % 64 == 0 && I >= 128 && I <= 8192 && ffn_lds_bytes(C, I) <= 160 * 1024;
}

int ffn_split_factor(int dtype, int C, int I) {
    // the estimator's block (C = 384, I = 1536): four workgroups per 128-row slab, 12 hidden tiles each
    if (!ffn_fused_supported(dtype, C, I) || C != 384 || I < 1024 || (I / 32) % 8) return 0;
    return 4;
}
bool ffn_split_valid(int dtype, int C, int I, int S) {
    return ffn_split_factor(dtype, C, I) > 1 && (S == 4 || S == 8 || S == 12 || S == 24) && (I / 32) % (2 * S) == 0;
}
// Few rows: more, shorter workgroups per slab (each still streams only ITS share of the weights, so the cost of a launch is one
// workgroup's prologue + its T = I/32/S hidden tiles + epilogue, whatever the number of slabs).  Measured on B sequences of 58 frames
// (tools/ffn_bench.py splitm, profiles/r03_ffn_split_small_m.txt; block = conv kernel + pointwise pair, us): 2..32 sequences
// (116..1856 rows) 12 ways 23-28 against 4 ways 28-31 and three launches 29-33; from 48 sequences on 4 ways wins (32 against 39
// and 40); 24 ways never did (its fold reads 24 partial sums per element), and ONE sequence ties with the three launches (20.9 /
// 21.2) — the launch is then a chain of fixed latencies (6 k cycles of prologue, 4 k of epilogue) around 12 k cycles of work.
// Which split a launch takes depends on its row count: results of different splits agree to rounding (16-bit partial sums), not
// bit for bit (include/stn.h).
int ffn_split_choose(int dtype, int C, int I, int64_t M) {
    if (ffn_split_factor(dtype, C, I) < 2) return 0;
    static const int force = [] { const char* e = stn::dev_env("STN_FFN_SPLIT_S"); return e ? atoi(e) : 0; }();  // A/B switch
    const int64_t nslab = (M + 127) / 128;
    // round 4 (profiles/r04_ffn_split_mid_m.txt, block = fold_dwconv_ln + K4-split, us): 8 ways fill the chip where 4 ways leave half of it idle —
    // 48 sequences 26.9 against 29.5 (4 ways) and 34.4 (12 ways: two rounds), 64 sequences 28.5 / 30.9 / 37.4, 32 sequences 25.4 / 28.0 / 26.1; from 96
    // sequences on 8 ways are two rounds (41.8 against 33.5) and 4 ways stay
    const int want = force ? force : nslab <= 12 ? 12 : nslab <= 32 ? 8 : 4;
    return ffn_split_valid(dtype, C, I, want) ? want : 4;
}

template <int C>
static void launch_ffn_t(hipStream_t s, int dtype, const FfnArgs& a) {
    const size_t lds = ffn_lds_bytes(C, a.I);
    static PerDeviceOnce attr_once;
    if (attr_once.need())
        stn_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_fused_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          160 * 1024), "hipFuncSetAttribute(ffn_fused)");
    const int nslab = (a.M + 127) / 128;
    const dim3 grid(a.split > 1 ? (unsigned)((nslab + 7) / 8 * 8 * a.split) : (unsigned)nslab);
#ifdef STN_FFN_VARIANTS
    static const int var = [] { const char* e = stn::dev_env("STN_FFN_VAR"); return e ? atoi(e) : 0; }();
    auto go = [&](auto kern) {
        stn_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "hipFuncSetAttribute(ffn_fused variant)");
        STN_KLAUNCH(kern, grid, dim3(256), lds, s, a);
    };
    if (var == 1) { go(&v_nodma::ffn_fused_kernel<C>); return; }
    if (var == 2) { go(&v_nogelu::ffn_fused_kernel<C>); return; }
    if (var == 3) { go(&v_earlyrd::ffn_fused_kernel<C>); return; }
    if (var == 4) { go(&v_bare::ffn_fused_kernel<C>); return; }
    if (var == 5) { go(&v_nord::ffn_fused_kernel<C>); return; }
    if (var == 6) { go(&v_nobar::ffn_fused_kernel<C>); return; }
    if (var == 7) { go(&v_nohand::ffn_fused_kernel<C>); return; }
    if (var == 8) { go(&v_mfmaonly::ffn_fused_kernel<C>); return; }
    if (var == 9) { go(&v_stamp::ffn_fused_kernel<C>); return; }
    if (var == 10) { go(&v_stamp_mfma::ffn_fused_kernel<C>); return; }
#endif
    if (dtype == F16) {
        static PerDeviceOnce attr16;
        if (attr16.need())
            stn_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&f16k::ffn_fused_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              160 * 1024), "hipFuncSetAttribute(ffn_fused f16)");
        STN_KLAUNCH((f16k::ffn_fused_kernel<C>), grid, dim3(256), lds, s, a);
        return;
    }
    STN_KLAUNCH((ffn_fused_kernel<C>), grid, dim3(256), lds, s, a);
}

void launch_ffn_fused(hipStream_t s, int dtype, int C, const FfnArgs& a) {
    if (a.M <= 0) return;
    if (!ffn_fused_supported(dtype, C, a.I)) throw std::invalid_argument("launch_ffn_fused: unsupported shape or dtype");
    if (a.split > 1 && (!ffn_split_valid(dtype, C, a.I, a.split) || !a.part || a.part_stride < ffn_split_rows(a.M) * C ||
                        (reinterpret_cast<uintptr_t>(a.part) & 15)))
        throw std::invalid_argument("launch_ffn_fused: hidden split needs a valid split (4, 8, 12 or 24 dividing I / 64) and a 16-byte aligned part buffer of [split][rows padded to 128][C]");
    if (a.split <= 1 && a.part) throw std::invalid_argument("launch_ffn_fused: part without split");
    if ((size_t)a.M * a.ldx * 2 >= 0x7FFFFFFFull || a.ldx % 8 || (reinterpret_cast<uintptr_t>(a.xn) & 15) ||
        (a.split <= 1 && (a.ldo % 4 || !a.x || (reinterpret_cast<uintptr_t>(a.x) & 15) || (a.rowvec && (a.rv_ld % 4 || (reinterpret_cast<uintptr_t>(a.rowvec) & 15))))))
        throw std::invalid_argument("launch_ffn_fused: operand alignment / size violates the kernel contract");
    if (C == 384) launch_ffn_t<384>(s, dtype, a);
    else launch_ffn_t<512>(s, dtype, a);
}

}  // namespace stn
