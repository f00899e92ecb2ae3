"""Kernel-level parity on a real MI355X, through the C ABI (stn_op_*), against the CPU oracle's
primitives (oracle/stn_ref.c) and plain fp32 numpy.  Tolerances are stated per test:
fp32 paths use the exact-fp32 MFMA (k-ordered FMA chain) -> 1e-5-level agreement;
bf16 paths round operands to 8 significant bits -> compared against an fp32 reference computed on
bf16-rounded operands (tight) and against the unrounded reference (loose)."""
import numpy as np
import pytest

from oracle.neural_ref import lib as reflib, randn as ref_randn
from supertonic_amd import binding
from supertonic_amd.arch import default_arch
from gpu_util import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = binding.Engine(0, "f32")
    e.load_synthetic(default_arch_small(), 7)
    return e


def default_arch_small():
    from supertonic_amd.arch import tiny_arch
    return tiny_arch()


def bf16_round(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32)


def gelu(x):
    from scipy.special import erf
    return 0.5 * x * (1 + erf(x / np.sqrt(2)))


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 144, 384), (49, 384, 144), (1000, 512, 2048), (7, 1, 128),
                                   (257, 130, 72),
                                   # tiny M, long K: the deterministic split-K path (8 / 8 / 6 splits)
                                   (49, 384, 1536), (128, 512, 2048), (64, 256, 1152),
                                   # tiny products (one thread per output, gemm_tiny_f32_kernel): the duration predictor's style K/V projection
                                   # and pooled layers, a single column, K not a multiple of 32
                                   (1024, 256, 16), (128, 128, 128), (128, 1, 128), (5, 7, 20), (64, 512, 256)])
def test_gemm_f32(eng, M, N, K):
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64).T + b
    got = eng.op_gemm(A, W, b, dtype="f32")
    mx, _ = rel_err(got, ref)
    assert mx < 2e-5, mx
    got = eng.op_gemm(A, W, b, act=binding.ACT_GELU, dtype="f32")
    mx, _ = rel_err(got, gelu(ref))
    assert mx < 2e-5, mx  # erf by Abramowitz-Stegun 7.1.26: |err| <= 1.5e-7


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 144, 384), (49, 384, 144), (1000, 512, 2048), (257, 130, 72)])
def test_gemm_bf16(eng, M, N, K):
    rng = np.random.default_rng(M * 3 + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    got = eng.op_gemm(A, W, None, dtype="bf16")
    ref_rounded = bf16_round(A).astype(np.float64) @ bf16_round(W).astype(np.float64).T
    mx, _ = rel_err(got, ref_rounded)
    assert mx < 2e-5, mx  # same rounded operands, fp32 accumulate
    mx, rms = rel_err(got, A.astype(np.float64) @ W.astype(np.float64).T)
    assert rms < 6e-3 and mx < 3e-2, (mx, rms)  # bf16 operand rounding: 2^-9 relative per operand


def f16_round(x):
    return np.ascontiguousarray(x, np.float32).astype(np.float16).astype(np.float32)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 144, 384), (49, 384, 144), (1000, 512, 2048), (257, 130, 72),
                                   (7436, 1536, 384), (600, 384, 1536)])
def test_gemm_f16(eng, M, N, K):
    """STN_DTYPE_F16 (BASELINE config 5, "fp16 MFMA linears"): IEEE-half operands, v_mfma_f32_32x32x16_f16, fp32 accumulate."""
    rng = np.random.default_rng(M * 5 + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    got = eng.op_gemm(A, W, b, dtype="f16")
    ref_rounded = f16_round(A).astype(np.float64) @ f16_round(W).astype(np.float64).T + b
    mx, _ = rel_err(got, ref_rounded)
    assert mx < 2e-5, mx  # same rounded operands, fp32 accumulate
    mx, rms = rel_err(got, A.astype(np.float64) @ W.astype(np.float64).T + b)
    assert rms < 8e-4 and mx < 4e-3, (mx, rms)  # half operand rounding: 2^-12 relative per operand, 8x tighter than bf16
    got = eng.op_gemm(A, W, b, act=binding.ACT_GELU, dtype="f16")
    mx, _ = rel_err(got, gelu(ref_rounded))
    assert mx < 2e-5, mx  # fp32 output: the erf form


def test_gemm_asymmetric_identity(eng):
    """A = I with an asymmetric W catches a swapped row/col accumulator map."""
    K = 128
    A = np.eye(K, dtype=np.float32)
    W = np.arange(K * K, dtype=np.float32).reshape(K, K) % 251
    for dt in ("f32", "bf16", "f16"):
        got = eng.op_gemm(A, W, None, dtype=dt)
        assert np.array_equal(got, W.T), dt


@pytest.mark.parametrize("C,k,dil,B,L", [(64, 5, 1, 3, 37), (384, 5, 8, 3, 37), (512, 7, 4, 3, 37), (256, 5, 2, 3, 37),
                                          (96, 7, 1, 3, 37),
                                          # comb kernel: R=4 (M >= 4096) and R=8 (M >= 32768), sequence tails not multiples of R*dil
                                          (384, 5, 8, 61, 78), (384, 5, 1, 60, 77), (512, 7, 4, 9, 471), (512, 7, 2, 70, 470),
                                          (512, 7, 1, 70, 469), (256, 5, 4, 200, 94)])
def test_dwconv_ln(eng, C, k, dil, B, L):
    rng = np.random.default_rng(C + k + dil)
    x = rng.standard_normal((B, L, C)).astype(np.float32)
    w = rng.standard_normal((C, k)).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    bt = (0.1 * rng.standard_normal(C)).astype(np.float32)
    h = np.empty_like(x)
    reflib().stnref_dwconv(x.reshape(-1, C), B, L, C, w, b, k, dil, h.reshape(-1, C))
    ref = np.empty_like(x)
    reflib().stnref_layernorm(h.reshape(-1, C), B * L, C, g, bt, 1e-6, ref.reshape(-1, C))
    got = eng.op_dwconv_ln(x, w, b, g, bt, dil, dtype="f32")
    mx, _ = rel_err(got, ref)
    assert mx < 2e-5, mx
    got = eng.op_dwconv_ln(x, w, b, g, bt, dil, dtype="bf16")
    # output rounded to bf16 (round-to-nearest-even): |err| <= 2^-8 |ref| elementwise (+ the fp32 noise floor)
    assert np.all(np.abs(got - ref) <= 2.0 ** -8 * np.abs(ref) + 2e-5)
    got = eng.op_dwconv_ln(x, w, b, g, bt, dil, dtype="f16")
    # output rounded to IEEE half (round-to-nearest-even): |err| <= 2^-11 |ref| elementwise (+ the fp32 noise floor)
    assert np.all(np.abs(got - ref) <= 2.0 ** -11 * np.abs(ref) + 2e-5)


@pytest.mark.parametrize("dh,H,Lq,Lk,rope", [(32, 2, 7, 11, -1), (64, 4, 70, 70, 0), (96, 4, 49, 62, 1), (48, 2, 33, 130, 1),
                                              (16, 2, 5, 6, 0), (64, 4, 200, 310, 0), (96, 2, 130, 50, -1), (32, 4, 129, 33, 1),
                                              # long contexts: keys stream through LDS in 128-key chunks (3 chunks here), partial last chunk
                                              (96, 4, 78, 311, 1), (96, 2, 140, 257, 0), (32, 2, 40, 129, -1),
                                              # a handful of keys, no rotation (fp32: attn_fewkeys_f32_kernel; the duration predictor's 8 style tokens)
                                              (64, 2, 70, 8, -1), (96, 1, 130, 16, -1), (16, 2, 5, 6, -1), (64, 2, 300, 1, -1)])
def test_attention(eng, dh, H, Lq, Lk, rope):
    rng = np.random.default_rng(dh + Lq + Lk)
    B, C = 2, dh * H
    q = rng.standard_normal((B, Lq, C)).astype(np.float32)
    k = rng.standard_normal((B, Lk, C)).astype(np.float32)
    v = rng.standard_normal((B, Lk, C)).astype(np.float32)
    qlen = np.array([Lq, max(1, Lq - 3)], np.int32)
    klen = np.array([Lk, max(1, Lk // 2)], np.int32)

    def rope_np(x, lens, mode):
        if mode < 0:
            return x
        Bn, Ln, _ = x.shape
        y = x.reshape(Bn, Ln, H, dh).astype(np.float64).copy()
        i = np.arange(dh // 2)
        inv = np.exp(-np.log(10000.0) * 2 * i / dh)
        for bb in range(Bn):
            pos = np.arange(Ln) * (10.0 / max(int(lens[bb]), 1)) if mode == 1 else np.arange(Ln).astype(np.float64)
            ang = pos[:, None] * inv[None, :]
            c, s = np.cos(ang)[:, None, :], np.sin(ang)[:, None, :]
            a0, a1 = y[bb, :, :, :dh // 2].copy(), y[bb, :, :, dh // 2:].copy()
            y[bb, :, :, :dh // 2] = a0 * c - a1 * s
            y[bb, :, :, dh // 2:] = a1 * c + a0 * s
        return y.reshape(Bn, Ln, C).astype(np.float32)

    qr, kr = rope_np(q, qlen, rope), rope_np(k, klen, rope)
    ref = np.empty_like(q)
    reflib().stnref_attention_core(qr.reshape(-1, C), kr.reshape(-1, C), v.reshape(-1, C), B, Lq, Lk, C, H,
                                   klen.ctypes.data, ref.reshape(-1, C))
    got = eng.op_attention(q, k, v, H, qlen, klen, rope, dtype="f32")
    mx, _ = rel_err(got, ref)
    assert mx < 5e-5, mx
    got = eng.op_attention(q, k, v, H, qlen, klen, rope, dtype="bf16")
    mx, rms = rel_err(got, ref)
    assert rms < 1.5e-2 and mx < 8e-2, (mx, rms)  # q,k,v,o rounded to bf16
    got = eng.op_attention(q, k, v, H, qlen, klen, rope, dtype="f16")
    mx, rms = rel_err(got, ref)
    assert rms < 2e-3 and mx < 1e-2, (mx, rms)  # q,k,v,p,o rounded to half: 8x tighter than bf16
    if rope >= 0:
        # keys rotated once by a separate pass (how the vector estimator treats its step-invariant text keys): same result
        got2 = eng.op_attention(q, k, v, H, qlen, klen, rope | 0x100, dtype="f32")
        mx, _ = rel_err(got2, ref)
        assert mx < 5e-5, mx
        got2 = eng.op_attention(q, k, v, H, qlen, klen, rope | 0x100, dtype="bf16")
        mx, rms = rel_err(got2, ref)
        assert rms < 1.5e-2 and mx < 8e-2, (mx, rms)
        got2 = eng.op_attention(q, k, v, H, qlen, klen, rope | 0x100, dtype="f16")
        mx, rms = rel_err(got2, ref)
        assert rms < 2e-3 and mx < 1e-2, (mx, rms)


def test_randn_matches_oracle_philox(eng):
    B, D, L = 3, 144, 50
    ids = np.array([5, 0, 123456789012], np.int64)
    ref = ref_randn(1234, B, D, L, ids)
    got = eng.op_randn(1234, B, D, L, ids, None)
    # identical Philox counters; libm vs device log/sin/cos differ by a few ulp
    assert np.abs(got - ref).max() < 2e-5
    ln = np.array([50, 20, 1], np.int32)
    got = eng.op_randn(1234, B, D, L, ids, ln)
    assert np.all(got[1, :, 20:] == 0) and np.all(got[2, :, 1:] == 0) and np.abs(got[1, :, :20] - ref[1, :, :20]).max() < 2e-5
