"""K4 — the fused pointwise pair of a ConvNeXt block (supertonic_amd/csrc/kernels_ffn.hip) on a real MI355X, through the
C ABI (stn_op_ffn), against (a) a float64 numpy restatement on the same bf16-rounded operands with the hidden activation
rounded to bf16 where the kernel rounds it, (b) the two tiled GEMM launches it replaces, (c) the CPU oracle end to end
(stage tests with the fused path switched on and off).  Stands in for the body of vocoder_ort_->Run / vector_est_ort_->Run
(/root/reference/cpp/helper.cpp:643-647, 668-672)."""
import numpy as np
import pytest

from supertonic_amd import binding
from supertonic_amd.arch import tiny_arch
from gpu_util import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = binding.Engine(0, "bf16")
    e.load_synthetic(tiny_arch(), 7)
    return e


def bf16_round(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32)


def gelu(x):
    from scipy.special import erf
    return 0.5 * x * (1 + erf(x / np.sqrt(2)))


def make(M, C, I, seed, nseq=0):
    rng = np.random.default_rng(seed)
    xn = rng.standard_normal((M, C)).astype(np.float32)
    W1 = (rng.standard_normal((I, C)) / np.sqrt(C)).astype(np.float32)
    W2 = (rng.standard_normal((C, I)) / np.sqrt(I)).astype(np.float32)
    b1 = (0.3 * rng.standard_normal(I)).astype(np.float32)
    b2 = (0.3 * rng.standard_normal(C)).astype(np.float32)
    gamma = (0.1 + 0.05 * rng.standard_normal(C)).astype(np.float32)
    x = rng.standard_normal((M, C)).astype(np.float32)
    rowvec = row_b = None
    if nseq:
        rowvec = rng.standard_normal((nseq, C)).astype(np.float32)
        row_b = np.sort(rng.integers(0, nseq, M)).astype(np.int32)
    return xn, W1, b1, W2, b2, gamma, x, rowvec, row_b


def ref64(xn, W1, b1, W2, b2, gamma, x, rowvec, row_b):
    h = bf16_round(xn).astype(np.float64) @ bf16_round(W1).astype(np.float64).T + b1
    g = bf16_round(gelu(h).astype(np.float32)).astype(np.float64)  # the hidden activation is a bf16 MFMA operand
    y = g @ bf16_round(W2).astype(np.float64).T + b2
    out = x + gamma * y
    if rowvec is not None:
        out = out + rowvec[row_b]
    return out


@pytest.mark.parametrize("M,C,I,nseq", [(128, 512, 2048, 0), (300, 512, 2048, 0), (7436, 384, 1536, 128), (1000, 384, 1536, 0),
                                         (4096, 512, 2048, 0), (31, 384, 192, 0)])
def test_ffn_fused_vs_float64(eng, M, C, I, nseq):
    ops = make(M, C, I, M + C + I, nseq)
    got = eng.op_ffn(*ops[:7], rowvec=ops[7], row_b=ops[8], fused=True)
    ref = ref64(*ops)
    upd = ref - ops[6]  # compare the UPDATE (the residual itself is copied through)
    mx, rms = rel_err(got - ops[6], upd)
    # bf16 GELU form (|err| <= 3e-4 absolute before rounding) + one bf16 rounding of the hidden activation that may fall on the
    # other side of a tie than the float64 reference's: a few 1e-3 of the update's rms
    assert rms < 3e-3 and mx < 3e-2, (mx, rms)
    assert np.all(np.isfinite(got))


@pytest.mark.parametrize("M,C,I,nseq", [(300, 512, 2048, 0), (7436, 384, 1536, 128), (2048, 512, 2048, 0)])
def test_ffn_fused_vs_two_launches(eng, M, C, I, nseq):
    """Same operands, same rounding points (xn, weights, hidden in bf16; fp32 accumulate): the fused kernel and the two GEMM
    launches differ by fp32 summation order and by hidden values that round to neighbouring bf16 numbers."""
    ops = make(M, C, I, 3 * M + C, nseq)
    a = eng.op_ffn(*ops[:7], rowvec=ops[7], row_b=ops[8], fused=True)
    b = eng.op_ffn(*ops[:7], rowvec=ops[7], row_b=ops[8], fused=False)
    mx, rms = rel_err(a - ops[6], b - ops[6])
    assert rms < 2e-3 and mx < 3e-2, (mx, rms)


def test_ffn_rows_do_not_depend_on_position(eng):
    """A row's result must not depend on where it sits in the launch (which wave, which workgroup): the exact trimmed dense
    vocoder and the packed layouts rely on it."""
    M, C, I = 700, 512, 2048
    ops = make(M, C, I, 5)
    full = eng.op_ffn(*ops[:7], fused=True)
    sel = np.r_[3:40, 129:300, 511:700]
    part = eng.op_ffn(ops[0][sel], *ops[1:6], ops[6][sel], fused=True)
    assert np.array_equal(full[sel], part)


def test_ffn_identity_weights_catch_layout_errors(eng):
    """W1 = [I_C; 0], W2 = W1^T scaled, asymmetric x: a wrong fragment order, k permutation or accumulator map cannot pass."""
    M, C, I = 200, 384, 384
    rng = np.random.default_rng(1)
    xn = bf16_round(rng.integers(-8, 9, (M, C)).astype(np.float32) / 4.0)
    W1 = np.eye(I, C, dtype=np.float32)
    P = rng.permutation(I)
    W2 = np.zeros((C, I), np.float32)
    W2[np.arange(C), P[:C]] = 1.0  # output channel n takes hidden unit P[n]
    b1 = np.zeros(I, np.float32)
    got = eng.op_ffn(xn, W1, b1, W2, None, None, np.zeros((M, C), np.float32), fused=True)
    ref = bf16_round(gelu(xn.astype(np.float64)).astype(np.float32))[:, P[:C]]
    assert np.max(np.abs(got - ref)) < 8e-3  # bf16 GELU form vs erf, then one bf16 rounding


# ---- the IEEE-half instantiation (f16 engines): same kernel text, v_mfma_f32_32x32x16_f16 / v_cvt_pk_f16_f32 ----------------

@pytest.fixture(scope="module")
def eng16():
    e = binding.Engine(0, "f16")
    e.load_synthetic(tiny_arch(), 7)
    return e


def f16_round(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


@pytest.mark.parametrize("M,C,I,nseq", [(300, 512, 2048, 0), (7436, 384, 1536, 128), (4096, 512, 2048, 0)])
def test_ffn_fused_f16_vs_float64_and_two_launches(eng16, M, C, I, nseq):
    ops = make(M, C, I, M + C + I + 1, nseq)
    xn, W1, b1, W2, b2, gamma, x, rowvec, row_b = ops
    got = eng16.op_ffn(*ops[:7], rowvec=rowvec, row_b=row_b, fused=True)
    h = f16_round(xn).astype(np.float64) @ f16_round(W1).astype(np.float64).T + b1
    y = f16_round(gelu(h).astype(np.float32)).astype(np.float64) @ f16_round(W2).astype(np.float64).T + b2
    ref = x + gamma * y + (rowvec[row_b] if rowvec is not None else 0.0)
    mx, rms = rel_err(got - x, ref - x)
    # the exp2-form GELU inside the blocks (|err| <= 5e-4 absolute) is what is left once the operands carry 11 bits
    assert rms < 1.5e-3 and mx < 1.5e-2 and np.all(np.isfinite(got)), (mx, rms)
    two = eng16.op_ffn(*ops[:7], rowvec=rowvec, row_b=row_b, fused=False)  # erf-form GELU, two launches
    mx2, rms2 = rel_err(got - x, two - x)
    assert rms2 < 1.5e-3 and mx2 < 1.5e-2, (mx2, rms2)


def test_ffn_f16_rows_do_not_depend_on_position(eng16):
    M, C, I = 700, 512, 2048
    ops = make(M, C, I, 6)
    full = eng16.op_ffn(*ops[:7], fused=True)
    sel = np.r_[3:40, 129:300, 511:700]
    part = eng16.op_ffn(ops[0][sel], *ops[1:6], ops[6][sel], fused=True)
    assert np.array_equal(full[sel], part)
