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
#include "kernels_dev.hpp"

#include <stdio.h>

namespace stn {

typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---- asm blocks --------------------------------------------------------------------------------------------------------
// Common: %[n0..n3] next fragments (outputs), %[ad] LDS byte address of this lane's 16 bytes in piece 0 of the stage the NEXT
// fragments come from, o0..o3 their piece offsets in bytes (immediates).
#define STN_RD4                                          \
    "ds_read_b128 %[n0], %[ad] offset:%[o0]\n\t"         \
    "ds_read_b128 %[n1], %[ad] offset:%[o1]\n\t"
#define STN_RD_2 "ds_read_b128 %[n2], %[ad] offset:%[o2]\n\t"
#define STN_RD_3 "ds_read_b128 %[n3], %[ad] offset:%[o3]\n\t"
// An MFMA result inside an asm statement is invisible to hipcc: whatever it places behind the statement (a live-range copy of
// the accumulator, an epilogue read) may read the destination before the 8-pass MFMA has written it (needs 11 wait states;
// found the hard way: a v_mov copy two instructions behind the last MFMA of a block lost that MFMA's contribution).  Every
// block therefore keeps >= 11 wait states between its last MFMA into an arch-VGPR accumulator and its end — the GELU tail where
// there is one, this pad where there is none (the last block of a phase-2 stage; its other blocks' tiles are four MFMAs older by then).
// tools/check_hazards.py walks the shipped code object for exactly this (tests/test_isa_cpu.py).
#define STN_MFMA_PAD "s_nop 7\n\ts_nop 2"
#define STN_MF1(f, x) "v_mfma_f32_32x32x16_bf16 %[acc], %[" #f "], %[" #x "], %[acc]\n\t"
// GELU of the pw1 epilogue (gelu_bf16_f: x / (1 + exp2(x * fma(x*x, -0.10294324, -2.30220819)))) on two accumulator
// elements, split into three groups that go behind successive MFMAs; every dependent pair has an instruction between it
// and its producer (the transcendental-result wait state of gfx950 is then satisfied without a stall)
#define STN_GELU_A(ha, hb, ta, tb)                                   \
    "v_mul_f32 %[" #ta "], %[" #ha "], %[" #ha "]\n\t"               \
    "v_mul_f32 %[" #tb "], %[" #hb "], %[" #hb "]\n\t"               \
    "v_fmamk_f32 %[" #ta "], %[" #ta "], 0xbdd2d3e8, %[cb]\n\t"      \
    "v_fmamk_f32 %[" #tb "], %[" #tb "], 0xbdd2d3e8, %[cb]\n\t"
#define STN_GELU_B(ha, hb, ta, tb)                                   \
    "v_mul_f32 %[" #ta "], %[" #ha "], %[" #ta "]\n\t"               \
    "v_mul_f32 %[" #tb "], %[" #hb "], %[" #tb "]\n\t"               \
    "v_exp_f32 %[" #ta "], %[" #ta "]\n\t"                           \
    "v_exp_f32 %[" #tb "], %[" #tb "]\n\t"
#define STN_GELU_C(ha, hb, ta, tb, gw)                               \
    "s_nop 0\n\t"                                                    \
    "v_add_f32 %[" #ta "], 1.0, %[" #ta "]\n\t"                      \
    "v_add_f32 %[" #tb "], 1.0, %[" #tb "]\n\t"                      \
    "v_rcp_f32 %[" #ta "], %[" #ta "]\n\t"                           \
    "v_rcp_f32 %[" #tb "], %[" #tb "]\n\t"                           \
    "s_nop 0\n\t"                                                    \
    "v_mul_f32 %[" #ta "], %[" #ha "], %[" #ta "]\n\t"               \
    "v_mul_f32 %[" #tb "], %[" #hb "], %[" #tb "]\n\t"               \
    "v_cvt_pk_bf16_f32 %[" #gw "], %[" #ta "], %[" #tb "]\n\t"

// phase-1 block: 4 MFMAs into the chained accumulator `acc` (arch VGPRs), no GELU
template <int O0, int O1, int O2, int O3>
__device__ __forceinline__ void p1_block_g0(f32x16& acc, const bf16x8 (&f)[4], bf16x8 (&n)[4], bf16x8 x0, bf16x8 x1, bf16x8 x2, bf16x8 x3,
                                            unsigned ad) {
    asm volatile(STN_RD4 STN_MF1(f0, x0) STN_RD_2 STN_MF1(f1, x1) STN_RD_3 STN_MF1(f2, x2) STN_MF1(f3, x3)
                 "s_waitcnt lgkmcnt(0)\n\t" STN_MFMA_PAD
                 : [n0] "=&v"(n[0]), [n1] "=&v"(n[1]), [n2] "=&v"(n[2]), [n3] "=&v"(n[3]), [acc] "+v"(acc)
                 : [f0] "v"(f[0]), [f1] "v"(f[1]), [f2] "v"(f[2]), [f3] "v"(f[3]), [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3),
                   [ad] "v"(ad), [o0] "i"(O0), [o1] "i"(O1), [o2] "i"(O2), [o3] "i"(O3)
                 : "memory");
}
// phase-1 block with the GELU of two elements (ha, hb) of the PREVIOUS tile -> one packed word
template <int O0, int O1, int O2, int O3>
__device__ __forceinline__ void p1_block_g2(f32x16& acc, const bf16x8 (&f)[4], bf16x8 (&n)[4], bf16x8 x0, bf16x8 x1, bf16x8 x2, bf16x8 x3,
                                            unsigned ad, float ha, float hb, float cb, unsigned& gw) {
    float ta, tb;
    asm volatile(STN_RD4 STN_MF1(f0, x0) STN_GELU_A(ha, hb, ta, tb) STN_RD_2 STN_MF1(f1, x1) STN_GELU_B(ha, hb, ta, tb) STN_RD_3
                 STN_MF1(f2, x2) STN_MF1(f3, x3) STN_GELU_C(ha, hb, ta, tb, gw)
                 "s_waitcnt lgkmcnt(0)\n\ts_nop 1"
                 : [n0] "=&v"(n[0]), [n1] "=&v"(n[1]), [n2] "=&v"(n[2]), [n3] "=&v"(n[3]), [acc] "+v"(acc), [ta] "=&v"(ta), [tb] "=&v"(tb),
                   [gw] "=&v"(gw)
                 : [f0] "v"(f[0]), [f1] "v"(f[1]), [f2] "v"(f[2]), [f3] "v"(f[3]), [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3),
                   [ad] "v"(ad), [ha] "v"(ha), [hb] "v"(hb), [cb] "v"(cb), [o0] "i"(O0), [o1] "i"(O1), [o2] "i"(O2), [o3] "i"(O3)
                 : "memory");
}
// ... of four elements -> two packed words (stages with fewer than eight blocks: C < 512)
template <int O0, int O1, int O2, int O3>
__device__ __forceinline__ void p1_block_g4(f32x16& acc, const bf16x8 (&f)[4], bf16x8 (&n)[4], bf16x8 x0, bf16x8 x1, bf16x8 x2, bf16x8 x3,
                                            unsigned ad, float ha, float hb, float hc, float hd, float cb, unsigned& gw0, unsigned& gw1) {
    float ta, tb;
    asm volatile(STN_RD4 STN_MF1(f0, x0) STN_GELU_A(ha, hb, ta, tb) STN_GELU_B(ha, hb, ta, tb) STN_RD_2 STN_MF1(f1, x1)
                 STN_GELU_C(ha, hb, ta, tb, gw0) STN_GELU_A(hc, hd, ta, tb) STN_RD_3 STN_MF1(f2, x2) STN_GELU_B(hc, hd, ta, tb)
                 STN_MF1(f3, x3) STN_GELU_C(hc, hd, ta, tb, gw1)
                 "s_waitcnt lgkmcnt(0)\n\ts_nop 1"
                 : [n0] "=&v"(n[0]), [n1] "=&v"(n[1]), [n2] "=&v"(n[2]), [n3] "=&v"(n[3]), [acc] "+v"(acc), [ta] "=&v"(ta), [tb] "=&v"(tb),
                   [gw0] "=&v"(gw0), [gw1] "=&v"(gw1)
                 : [f0] "v"(f[0]), [f1] "v"(f[1]), [f2] "v"(f[2]), [f3] "v"(f[3]), [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3),
                   [ad] "v"(ad), [ha] "v"(ha), [hb] "v"(hb), [hc] "v"(hc), [hd] "v"(hd), [cb] "v"(cb), [o0] "i"(O0), [o1] "i"(O1),
                   [o2] "i"(O2), [o3] "i"(O3)
                 : "memory");
}
// phase-2 block: pieces (nt, s) = (2b, 0), (2b, 1), (2b+1, 0), (2b+1, 1) -> two accumulator tiles in the AGPR file
template <int O0, int O1, int O2, int O3>
__device__ __forceinline__ void p2_block(f32x16& ya, f32x16& yb, const bf16x8 (&f)[4], bf16x8 (&n)[4], bf16x8 g0, bf16x8 g1, unsigned ad) {
    asm volatile(STN_RD4
                 "v_mfma_f32_32x32x16_bf16 %[ya], %[f0], %[g0], %[ya]\n\t" STN_RD_2
                 "v_mfma_f32_32x32x16_bf16 %[yb], %[f2], %[g0], %[yb]\n\t" STN_RD_3
                 "v_mfma_f32_32x32x16_bf16 %[ya], %[f1], %[g1], %[ya]\n\t"
                 "v_mfma_f32_32x32x16_bf16 %[yb], %[f3], %[g1], %[yb]\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : [n0] "=&v"(n[0]), [n1] "=&v"(n[1]), [n2] "=&v"(n[2]), [n3] "=&v"(n[3]), [ya] "+a"(ya), [yb] "+a"(yb)
                 : [f0] "v"(f[0]), [f1] "v"(f[1]), [f2] "v"(f[2]), [f3] "v"(f[3]), [g0] "v"(g0), [g1] "v"(g1), [ad] "v"(ad), [o0] "i"(O0),
                   [o1] "i"(O1), [o2] "i"(O2), [o3] "i"(O3)
                 : "memory");
}
// ... the last block of a phase-2 stage also fetches b1 of the hidden tile whose phase 1 comes next (4 x 16 bytes at `bad`)
template <int O0, int O1, int O2, int O3>
__device__ __forceinline__ void p2_block_bias(f32x16& ya, f32x16& yb, const bf16x8 (&f)[4], bf16x8 (&n)[4], bf16x8 g0, bf16x8 g1, unsigned ad,
                                              unsigned bad, f32x4 (&bq)[4]) {
    asm volatile(STN_RD4
                 "v_mfma_f32_32x32x16_bf16 %[ya], %[f0], %[g0], %[ya]\n\t" STN_RD_2
                 "ds_read_b128 %[b0], %[bad]\n\t"
                 "ds_read_b128 %[b1], %[bad] offset:32\n\t"
                 "v_mfma_f32_32x32x16_bf16 %[yb], %[f2], %[g0], %[yb]\n\t" STN_RD_3
                 "ds_read_b128 %[b2], %[bad] offset:64\n\t"
                 "ds_read_b128 %[b3], %[bad] offset:96\n\t"
                 "v_mfma_f32_32x32x16_bf16 %[ya], %[f1], %[g1], %[ya]\n\t"
                 "v_mfma_f32_32x32x16_bf16 %[yb], %[f3], %[g1], %[yb]\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t" STN_MFMA_PAD  /* the stage's last MFMAs: whatever hipcc puts behind the stage may read them */
                 : [n0] "=&v"(n[0]), [n1] "=&v"(n[1]), [n2] "=&v"(n[2]), [n3] "=&v"(n[3]), [ya] "+a"(ya), [yb] "+a"(yb), [b0] "=&v"(bq[0]),
                   [b1] "=&v"(bq[1]), [b2] "=&v"(bq[2]), [b3] "=&v"(bq[3])
                 : [f0] "v"(f[0]), [f1] "v"(f[1]), [f2] "v"(f[2]), [f3] "v"(f[3]), [g0] "v"(g0), [g1] "v"(g1), [ad] "v"(ad), [bad] "v"(bad),
                   [o0] "i"(O0), [o1] "i"(O1), [o2] "i"(O2), [o3] "i"(O3)
                 : "memory");
}
// four fragments (pieces 0..3 at `ad`): the pipeline's start
__device__ __forceinline__ void prime_frags(bf16x8 (&n)[4], unsigned ad) {
    asm volatile("ds_read_b128 %[n0], %[ad]\n\tds_read_b128 %[n1], %[ad] offset:1024\n\tds_read_b128 %[n2], %[ad] offset:2048\n\t"
                 "ds_read_b128 %[n3], %[ad] offset:3072\n\ts_waitcnt lgkmcnt(0)"
                 : [n0] "=&v"(n[0]), [n1] "=&v"(n[1]), [n2] "=&v"(n[2]), [n3] "=&v"(n[3])
                 : [ad] "v"(ad)
                 : "memory");
}
__device__ __forceinline__ void read_bias(f32x4 (&bq)[4], unsigned bad) {
    asm volatile("ds_read_b128 %[b0], %[bad]\n\tds_read_b128 %[b1], %[bad] offset:32\n\tds_read_b128 %[b2], %[bad] offset:64\n\t"
                 "ds_read_b128 %[b3], %[bad] offset:96\n\ts_waitcnt lgkmcnt(0)"
                 : [b0] "=&v"(bq[0]), [b1] "=&v"(bq[1]), [b2] "=&v"(bq[2]), [b3] "=&v"(bq[3])
                 : [bad] "v"(bad)
                 : "memory");
}
__device__ __forceinline__ void bias_to_acc(const f32x4 (&bq)[4], f32x16& h) {  // register 4q + j of lane half hf = unit 8q + 4hf + j
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) h[4 * q + j] = bq[q][j];
}

// Stage sg of the sequence W1(0), W1(1), W2(0), W1(2), W2(1), ..., W1(T-1), W2(T-2), W2(T-1): this wave's PER KiB pieces
// (plain functions with by-value arguments, not lambdas: nested by-reference closures end up in scratch)
template <int SB, int PER>
__device__ __forceinline__ void ffn_issue(int sg, int T, __amdgpu_buffer_rsrc_t rs1, __amdgpu_buffer_rsrc_t rs2, unsigned char* smem, int wave,
                                          unsigned voff) {
    const int NS = 2 * T;
    int kind, tile;
    if (sg == 0) { kind = 0; tile = 0; }
    else if (sg == NS - 1) { kind = 1; tile = T - 1; }
    else { const int u = sg - 1; kind = u & 1; tile = (u >> 1) + (kind ? 0 : 1); }
    unsigned char* dst = smem + (sg & 3) * SB + wave * (PER * 1024);
    const int so = tile * SB;
    if (kind) {
#pragma unroll
        for (int j = 0; j < PER; ++j) dma16(rs2, dst + j * 1024, voff + j * 1024, so);
    } else {
#pragma unroll
        for (int j = 0; j < PER; ++j) dma16(rs1, dst + j * 1024, voff + j * 1024, so);
    }
}
// Hand-over of stage q (q >= 1), executed in the middle of stage q-1: this wave's pieces of stage q have landed (the stages
// issued after it, at most two, may still be in flight), every wave is past stage q-2, whose buffer takes stage q+2.
template <int SB, int PER>
__device__ __forceinline__ void ffn_handover(int q, int T, __amdgpu_buffer_rsrc_t rs1, __amdgpu_buffer_rsrc_t rs2, unsigned char* smem, int wave,
                                             unsigned voff, int dbg = 0) {
    const int NS = 2 * T;
    if (dbg & 1) wait_vm<0>();
    const int issued = q <= 1 ? 3 : q + 1;                  // stages 0..3 go out in the prologue, stage q+2 at hand-over q
    const int ahead = (issued < NS - 1 ? issued : NS - 1) - q;
    if (ahead >= 2) wait_vm<2 * PER>();
    else if (ahead == 1) wait_vm<PER>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (q >= 2 && q + 2 < NS) ffn_issue<SB, PER>(q + 2, T, rs1, rs2, smem, wave, voff);
}

// One stage = NB blocks of four MFMAs.  `f` holds the fragments of block 0 on entry and of the NEXT stage's block 0 on exit;
// the hand-over of the next stage runs before block NB/2.
template <int C, bool GELU>
__device__ __forceinline__ void ffn_stage_p1(f32x16& acc, const f32x16& hprev, bf16x8 (&g)[2], bf16x8 (&f)[4], const bf16x8 (&xf)[C / 16],
                                             unsigned ad_cur, unsigned ring_ad, int q_next, int T, __amdgpu_buffer_rsrc_t rs1,
                                             __amdgpu_buffer_rsrc_t rs2, unsigned char* smem, int wave, unsigned voff, float cb, int dbg) {
    constexpr int NB = C / 64, SB = C * 64, PER = SB / 4096;
    static_assert(NB == 4 || NB == 6 || NB == 8, "blocks per stage");
    bf16x8 n[4];
    unsigned gw[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    const unsigned ad_nxt = ring_ad + (unsigned)(q_next & 3) * SB;
#define STN_XF(b) xf[4 * (b)], xf[4 * (b) + 1], xf[4 * (b) + 2], xf[4 * (b) + 3]
#define STN_P1(b, F, N, AD, OB)                                                                                                           \
    do {                                                                                                                                  \
        if ((b) == NB / 2 && q_next < 2 * T) ffn_handover<SB, PER>(q_next, T, rs1, rs2, smem, wave, voff, dbg);                                \
        if constexpr (!GELU) p1_block_g0<OB, OB + 1024, OB + 2048, OB + 3072>(acc, F, N, STN_XF(b), AD);                                  \
        else if constexpr (NB == 8) p1_block_g2<OB, OB + 1024, OB + 2048, OB + 3072>(acc, F, N, STN_XF(b), AD, hprev[(2 * (b)) & 15], hprev[(2 * (b) + 1) & 15], cb, gw[(b) & 7]); \
        else if constexpr (NB == 4 || (b) < 2) p1_block_g4<OB, OB + 1024, OB + 2048, OB + 3072>(acc, F, N, STN_XF(b), AD, hprev[(4 * (b)) & 15], hprev[(4 * (b) + 1) & 15], hprev[(4 * (b) + 2) & 15], hprev[(4 * (b) + 3) & 15], cb, gw[(2 * (b)) & 7], gw[(2 * (b) + 1) & 7]); \
        else p1_block_g2<OB, OB + 1024, OB + 2048, OB + 3072>(acc, F, N, STN_XF(b), AD, hprev[(4 + 2 * (b)) & 15], hprev[(5 + 2 * (b)) & 15], cb, gw[(2 + (b)) & 7]); \
    } while (0)
    // blocks alternate the two fragment sets; block b reads pieces 4(b+1).. of this stage, the last one pieces 0..3 of the next
    STN_P1(0, f, n, ad_cur, 4096);
    STN_P1(1, n, f, ad_cur, 8192);
    STN_P1(2, f, n, ad_cur, 12288);
    if constexpr (NB == 4) { STN_P1(3, n, f, ad_nxt, 0); }
    else {
        STN_P1(3, n, f, ad_cur, 16384);
        STN_P1(4, f, n, ad_cur, 20480);
        if constexpr (NB == 6) { STN_P1(5, n, f, ad_nxt, 0); }
        else {
            STN_P1(5, n, f, ad_cur, 24576);
            STN_P1(6, f, n, ad_cur, 28672);
            STN_P1(7, n, f, ad_nxt, 0);
        }
    }
#undef STN_P1
#undef STN_XF
    if constexpr (GELU) {
        const u32x4 v0 = {gw[0], gw[1], gw[2], gw[3]}, v1 = {gw[4], gw[5], gw[6], gw[7]};
        g[0] = __builtin_bit_cast(bf16x8, v0);
        g[1] = __builtin_bit_cast(bf16x8, v1);
    }
}
template <int C>
__device__ __forceinline__ void ffn_stage_p2(f32x16 (&yacc)[C / 32], const bf16x8 (&g)[2], bf16x8 (&f)[4], unsigned ad_cur, unsigned ring_ad,
                                             int q_next, int T, __amdgpu_buffer_rsrc_t rs1, __amdgpu_buffer_rsrc_t rs2, unsigned char* smem,
                                             int wave, unsigned voff, unsigned bad, f32x4 (&bq)[4], int dbg) {
    constexpr int NB = C / 64, SB = C * 64, PER = SB / 4096;
    bf16x8 n[4];
    const unsigned ad_nxt = ring_ad + (unsigned)(q_next & 3) * SB;
#define STN_P2(b, F, N, AD, OB)                                                                                    \
    do {                                                                                                           \
        if ((b) == NB / 2 && q_next < 2 * T) ffn_handover<SB, PER>(q_next, T, rs1, rs2, smem, wave, voff, dbg);         \
        if constexpr ((b) == NB - 1) p2_block_bias<OB, OB + 1024, OB + 2048, OB + 3072>(yacc[2 * (b)], yacc[2 * (b) + 1], F, N, g[0], g[1], AD, bad, bq); \
        else p2_block<OB, OB + 1024, OB + 2048, OB + 3072>(yacc[2 * (b)], yacc[2 * (b) + 1], F, N, g[0], g[1], AD); \
    } while (0)
    STN_P2(0, f, n, ad_cur, 4096);
    STN_P2(1, n, f, ad_cur, 8192);
    STN_P2(2, f, n, ad_cur, 12288);
    if constexpr (NB == 4) { STN_P2(3, n, f, ad_nxt, 0); }
    else {
        STN_P2(3, n, f, ad_cur, 16384);
        STN_P2(4, f, n, ad_cur, 20480);
        if constexpr (NB == 6) { STN_P2(5, n, f, ad_nxt, 0); }
        else {
            STN_P2(5, n, f, ad_cur, 24576);
            STN_P2(6, f, n, ad_cur, 28672);
            STN_P2(7, n, f, ad_nxt, 0);
        }
    }
#undef STN_P2
}

// GELU of a whole tile outside a phase-1 stage (the last tile): same arithmetic, compiler-scheduled
__device__ __forceinline__ void ffn_gelu_pack(const f32x16& h, bf16x8 (&g)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        unsigned w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            w[j] = (unsigned)cvt16<false>(gelu_bf16_f(h[8 * s + 2 * j])) | ((unsigned)cvt16<false>(gelu_bf16_f(h[8 * s + 2 * j + 1])) << 16);
        const u32x4 v = {w[0], w[1], w[2], w[3]};
        g[s] = __builtin_bit_cast(bf16x8, v);
    }
}

template <int C>
__global__ __launch_bounds__(256, 1) void ffn_fused_kernel(FfnArgs p) {
    constexpr int NKS = C / 16;       // phase-1 k-steps = KiB pieces of a W1 stage
    constexpr int NT2 = C / 32;       // output-channel tiles
    constexpr int SB = C * 64;        // bytes per stage (W1 tile: 32 x C x 2; W2 tile: C x 32 x 2)
    constexpr int PER = SB / 4096;    // DMA pieces per wave and stage
    constexpr int NST = 4;
    static_assert(C % 128 == 0 && C >= 256 && C <= 512, "shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // ring + b2 + gamma + b1: the only LDS object
    float* const b2s = reinterpret_cast<float*>(smem + NST * SB);
    float* const gms = b2s + C;
    float* const b1s = gms + C;
    const int M = p.M, I = p.I, ldx = p.ldx, ldo = p.ldo, rv_ld = p.rv_ld, Lseq = p.L;
    const void* const xn = p.xn; const void* const w1f = p.w1f; const void* const w2f = p.w2f;
    const float* const b1 = p.b1; const float* const b2 = p.b2; const float* const gamma = p.gamma;
    float* const x = p.x; const int* const row_b = p.row_b; const float* const rowvec = p.rowvec; const int* const len = p.len;
    unsigned long long* const ts = p.ts;
    const int dbg = p.dbg;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int T = I >> 5, NS = 2 * T;  // T is even (I % 64 == 0: the tile loop is unrolled by two)
    const int m = (int)blockIdx.x * 128 + wave * 32 + lr;  // this lane's row
    unsigned long long t_in = 0, t_first = 0, t_loop = 0;
    if (ts) t_in = __builtin_readcyclecounter();

    // this wave's rows of xn as phase-1 B fragments: lane (r, hf), k-step s holds xn[r][16s + 8hf .. +8]; rows >= M read as zeros
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(xn, (size_t)M * ldx * 2);
    bf16x8 xf[NKS];
    {
        const unsigned xo = m < M ? (unsigned)m * (unsigned)ldx * 2u + (unsigned)lh * 16u : OOB;
#pragma unroll
        for (int s = 0; s < NKS; ++s) xf[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsx, xo, s * 32, 0));
    }
    // biases and layer scale into LDS (before any DMA is in flight: a plain barrier)
    for (int i = tid; i < I; i += 256) b1s[i] = b1[i];
    for (int i = tid; i < C; i += 256) { b2s[i] = b2 ? b2[i] : 0.f; gms[i] = gamma ? gamma[i] : 1.f; }
    __syncthreads();
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned ring_ad = lds0 + (unsigned)lane * 16u;                                   // this lane's 16 bytes of piece 0, stage buffer 0
    const unsigned bias_ad = lds0 + (unsigned)(NST * SB + 2 * C * 4) + (unsigned)lh * 16u;  // b1s + 4hf floats; tile t at + t*128 bytes
    f32x16 hA, hB;
    f32x4 bq[4];
    read_bias(bq, bias_ad);
    bias_to_acc(bq, hA);
    read_bias(bq, bias_ad + 128u);
    bias_to_acc(bq, hB);

    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(w1f, (size_t)I * C * 2), rs2 = make_rsrc(w2f, (size_t)I * C * 2);
    const unsigned voff = (unsigned)(wave * PER * 1024 + lane * 16);
#pragma unroll
    for (int st = 0; st < NST; ++st)
        if (st < NS) ffn_issue<SB, PER>(st, T, rs1, rs2, smem, wave, voff);

    f32x16 yacc[NT2];
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) yacc[nt][i] = 0.f;
    float cb = __builtin_bit_cast(float, 0xc0135761u);  // -2.30220819f
    asm volatile("" : "+v"(cb));

    if (dbg & 2) {  // diagnostics: the ring as the prologue filled it -> x (raw words), nothing else
        wait_vm<0>();
        __syncthreads();
        if (blockIdx.x == 0)
            for (int i = tid; i < NST * SB / 4; i += 256) reinterpret_cast<unsigned*>(x)[i] = reinterpret_cast<unsigned*>(smem)[i];
        return;
    }
    // stage 0: wait for it (three younger stages and nothing else are in flight), prime the fragment pipeline
    wait_vm<3 * PER>();
    __builtin_amdgcn_s_barrier();
    if (ts) t_first = __builtin_readcyclecounter();
    bf16x8 f[4], g[2];
    prime_frags(f, ring_ad);
    int q = 0;  // the stage being computed
    ffn_stage_p1<C, false>(hA, hA, g, f, xf, ring_ad + (unsigned)(q & 3) * SB, ring_ad, q + 1, T, rs1, rs2, smem, wave, voff, cb, dbg);
    ++q;
    for (int t = 0; t < T; t += 2) {
        // phase 1 of tile t+1 into hB with the GELU of tile t (hA); phase 2 of tile t; then the same with the roles swapped
        ffn_stage_p1<C, true>(hB, hA, g, f, xf, ring_ad + (unsigned)(q & 3) * SB, ring_ad, q + 1, T, rs1, rs2, smem, wave, voff, cb, dbg);
        ++q;
        ffn_stage_p2<C>(yacc, g, f, ring_ad + (unsigned)(q & 3) * SB, ring_ad, q + 1, T, rs1, rs2, smem, wave, voff,
                        bias_ad + (unsigned)(t + 2 < T ? t + 2 : 0) * 128u, bq, dbg);
        ++q;
        if (t + 2 < T) {
            bias_to_acc(bq, hA);
            ffn_stage_p1<C, true>(hA, hB, g, f, xf, ring_ad + (unsigned)(q & 3) * SB, ring_ad, q + 1, T, rs1, rs2, smem, wave, voff, cb, dbg);
            ++q;
        } else {
            asm volatile("s_nop 7\n\ts_nop 4" : "+v"(hB));  // MFMA results of the last phase 1 -> VALU readers (12 wait states)
            ffn_gelu_pack(hB, g);
        }
        ffn_stage_p2<C>(yacc, g, f, ring_ad + (unsigned)(q & 3) * SB, ring_ad, q + 1, T, rs1, rs2, smem, wave, voff,
                        bias_ad + (unsigned)(t + 3 < T ? t + 3 : 0) * 128u, bq, dbg);
        ++q;
        bias_to_acc(bq, hB);
    }
    if (ts) t_loop = __builtin_readcyclecounter();

    // ---- epilogue: residual update from the accumulators, 16 bytes per access ---------------------------------------------
    // yacc[nt][4q + j] is channel 32nt + 8q + 4hf + j of row r: a lane owns 4 consecutive channels per register quad
    const bool row_ok = m < M;
    int bsel = 0;
    float keep = 1.f;
    if (row_ok) {
        if (row_b) bsel = row_b[m];
        else if (len || rowvec) { bsel = m / Lseq; if (len && m - bsel * Lseq >= len[bsel]) keep = 0.f; }
    }
    float* xr = x + (size_t)(row_ok ? m : 0) * ldo;
    const float* rv = rowvec ? rowvec + (size_t)bsel * rv_ld : nullptr;
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) {
        float4 r[4], tv[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int n = 32 * nt + 8 * qq + 4 * lh;
            r[qq] = make_float4(0.f, 0.f, 0.f, 0.f); tv[qq] = r[qq];
            if (row_ok) { r[qq] = *reinterpret_cast<const float4*>(xr + n); if (rv) tv[qq] = *reinterpret_cast<const float4*>(rv + n); }
        }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int n = 32 * nt + 8 * qq + 4 * lh;
            const float4 bb = *reinterpret_cast<const float4*>(b2s + n), gm = *reinterpret_cast<const float4*>(gms + n);
            float4 o;
            o.x = (r[qq].x + gm.x * (yacc[nt][4 * qq] + bb.x) + tv[qq].x) * keep;
            o.y = (r[qq].y + gm.y * (yacc[nt][4 * qq + 1] + bb.y) + tv[qq].y) * keep;
            o.z = (r[qq].z + gm.z * (yacc[nt][4 * qq + 2] + bb.z) + tv[qq].z) * keep;
            o.w = (r[qq].w + gm.w * (yacc[nt][4 * qq + 3] + bb.w) + tv[qq].w) * keep;
            if (row_ok) *reinterpret_cast<float4*>(xr + n) = o;
        }
    }
    if (ts && tid == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        unsigned long long* tp = ts + (size_t)blockIdx.x * 4;
        tp[0] = t_in; tp[1] = t_first; tp[2] = t_loop; tp[3] = __builtin_readcyclecounter();
    }
}

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

bool ffn_fused_supported(int dtype, int C, int I) {
    // bf16 only: the GELU inside the asm blocks is the bf16 form of the pw1 epilogue (half keeps the erf form and stays unfused)
    return dtype == BF16 && (C == 256 || C == 384 || C == 512) && I % 64 == 0 && I >= 128 && I <= 8192;
}

template <int C>
static void launch_ffn_t(hipStream_t s, const FfnArgs& a) {
    const size_t lds = (size_t)4 * C * 64 + (size_t)(a.I + 2 * C) * 4;
    static PerDeviceOnce attr_once;
    if (attr_once.need())
        stn_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_fused_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          160 * 1024), "hipFuncSetAttribute(ffn_fused)");
    STN_KLAUNCH((ffn_fused_kernel<C>), dim3((unsigned)((a.M + 127) / 128)), dim3(256), lds, s, a);
}

void launch_ffn_fused(hipStream_t s, int dtype, int C, const FfnArgs& a) {
    if (a.M <= 0) return;
    if (!ffn_fused_supported(dtype, C, a.I)) throw std::invalid_argument("launch_ffn_fused: unsupported shape or dtype");
    if ((size_t)a.M * a.ldx * 2 >= 0x7FFFFFFFull || a.ldx % 8 || a.ldo % 4 || (reinterpret_cast<uintptr_t>(a.xn) & 15) ||
        (reinterpret_cast<uintptr_t>(a.x) & 15) || (a.rowvec && (a.rv_ld % 4 || (reinterpret_cast<uintptr_t>(a.rowvec) & 15))))
        throw std::invalid_argument("launch_ffn_fused: operand alignment / size violates the kernel contract");
    FfnArgs b = a;
    static const int dbg_env = [] { const char* e = getenv("STN_FFN_DBG"); return e ? atoi(e) : 0; }();
    b.dbg |= dbg_env;
    switch (C) {
        case 256: launch_ffn_t<256>(s, b); break;
        case 384: launch_ffn_t<384>(s, b); break;
        default: launch_ffn_t<512>(s, b); break;
    }
}

}  // namespace stn
