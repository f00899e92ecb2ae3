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


# ---- K4-split: the hidden dimension cut over four workgroups per slab, 16-bit partial sums, folded by the next reader of x ----------

def ref64_split(xn, W1, b1, W2, b2, gamma, x, rowvec, row_b, S=4, rnd=bf16_round):
    """float64 restatement with the kernel's rounding points: xn, weights and the GELU'd hidden activation in the 16-bit format, each
    hidden quarter's contribution rounded to 16 bits, then the fold in fp32 in the kernels' order."""
    h = rnd(xn).astype(np.float64) @ rnd(W1).astype(np.float64).T + b1
    g = rnd(gelu(h).astype(np.float32)).astype(np.float64)
    I = W1.shape[0]
    q = I // S
    w2 = rnd(W2).astype(np.float64)
    parts = [rnd((g[:, s * q:(s + 1) * q] @ w2[:, s * q:(s + 1) * q].T).astype(np.float32)) for s in range(S)]
    y = parts[0]
    for p in parts[1:]:
        y = (y + p).astype(np.float32)
    out = (x + gamma * (y + b2).astype(np.float32)).astype(np.float32)
    if rowvec is not None:
        out = out + rowvec[row_b]
    return out


@pytest.mark.parametrize("M,nseq", [(7436, 128), (3000, 50), (1000, 0), (129, 0), (31, 3)])
def test_ffn_split_vs_float64_and_two_launches(eng, M, nseq):
    C, I = 384, 1536
    ops = make(M, C, I, 7 * M + 1, nseq)
    got = eng.op_ffn(*ops[:7], rowvec=ops[7], row_b=ops[8], fused=2)
    assert np.all(np.isfinite(got))
    x = ops[6]
    S = 12 if (M + 127) // 128 <= 12 else 8 if (M + 127) // 128 <= 32 else 4  # ffn_split_choose
    mx, rms = rel_err(got - x, ref64_split(*ops, S=S) - x)
    assert rms < 3e-3 and mx < 3e-2, (mx, rms)
    # against the unsplit float64 reference and the two launches: the 16-bit partial sums add ~2^-9 of a quarter's contribution
    mx, rms = rel_err(got - x, ref64(*ops) - x)
    assert rms < 6e-3 and mx < 5e-2, (mx, rms)
    two = eng.op_ffn(*ops[:7], rowvec=ops[7], row_b=ops[8], fused=0)
    mx, rms = rel_err(got - x, two - x)
    assert rms < 6e-3 and mx < 5e-2, (mx, rms)


def test_ffn_split_rows_do_not_depend_on_position_or_row_count(eng):
    """Within one split regime (here 12 ways: up to 12 slabs of 128 rows) a row gives the same bits alone, in another slab, in another
    launch size; across the boundaries (8 ways from 13 slabs on, 4 ways from 33) the results agree to rounding (next test)."""
    M, C, I = 700, 384, 1536
    ops = make(M, C, I, 9)
    full = eng.op_ffn(*ops[:7], fused=2)
    sel = np.r_[3:40, 129:300, 511:700]
    part = eng.op_ffn(ops[0][sel], *ops[1:6], ops[6][sel], fused=2)
    assert np.array_equal(full[sel], part)
    one = eng.op_ffn(ops[0][5:6], *ops[1:6], ops[6][5:6], fused=2)
    assert np.array_equal(full[5:6], one)
    # ... and within the 8-way regime (13 .. 32 slabs)
    M = 3000
    ops = make(M, C, I, 10)
    full = eng.op_ffn(*ops[:7], fused=2)
    part = eng.op_ffn(ops[0][:2000], *ops[1:6], ops[6][:2000], fused=2)
    assert np.array_equal(full[:2000], part)


def test_ffn_split_regimes_agree_to_rounding(eng):
    """1 536 rows run 12 ways, 1 664 rows 8 ways, 4 224 rows 4 ways: the shared rows differ only by the rounding of the 16-bit partial sums."""
    C, I = 384, 1536
    ops = make(4224, C, I, 11)
    runs = {n: eng.op_ffn(ops[0][:n], *ops[1:6], ops[6][:n], fused=2) for n in (1536, 1664, 4224)}
    x = ops[6][:1536]
    for a, b in ((1536, 1664), (1664, 4224), (1536, 4224)):
        assert not np.array_equal(runs[a][:1536], runs[b][:1536])
        mx, rms = rel_err(runs[a][:1536] - x, runs[b][:1536] - x)
        assert rms < 4e-3 and mx < 4e-2, (a, b, mx, rms)
    for n, got in runs.items():
        sub = (ops[0][:n], *ops[1:6], ops[6][:n], None, None)
        mx, rms = rel_err(got - ops[6][:n], ref64(*sub) - ops[6][:n])
        assert rms < 6e-3 and mx < 5e-2, (n, mx, rms)


def test_ffn_split_f16(eng16):
    M, C, I = 1500, 384, 1536
    ops = make(M, C, I, 77, 16)
    got = eng16.op_ffn(*ops[:7], rowvec=ops[7], row_b=ops[8], fused=2)
    x = ops[6]
    mx, rms = rel_err(got - x, ref64_split(*ops, S=12, rnd=f16_round) - x)
    assert rms < 1.5e-3 and mx < 1.5e-2 and np.all(np.isfinite(got)), (mx, rms)


def _fold_dwconv_ref(seqlen, x, part16, b2, gamma, rowvec, w, bias, g, b, k, dil, eps=1e-6):
    S, M, C = part16.shape
    y = part16[0]
    for s in range(1, S):
        y = (y + part16[s]).astype(np.float32)
    off = np.r_[0, np.cumsum(seqlen)]
    seq = np.repeat(np.arange(len(seqlen)), seqlen)
    xo = (x + gamma * (y + b2).astype(np.float32)).astype(np.float32)
    if rowvec is not None:
        xo = (xo + rowvec[seq]).astype(np.float32)
    out = np.zeros((M, C), np.float64)
    half = (k - 1) // 2
    for bi, n in enumerate(seqlen):
        xs = xo[off[bi]:off[bi] + n].astype(np.float64)
        acc = np.tile(bias.astype(np.float64), (n, 1))
        for j in range(k):
            sh = (j - half) * dil
            lo, hi = max(0, -sh), min(n, n - sh)
            if hi > lo:
                acc[lo:hi] += w[:, j] * xs[lo + sh:hi + sh]
        mu = acc.mean(1, keepdims=True)
        var = ((acc - mu) ** 2).mean(1, keepdims=True)
        out[off[bi]:off[bi] + n] = (acc - mu) / np.sqrt(var + eps) * g + b
    return xo, out


@pytest.mark.parametrize("k,dil,C,S", [(5, 1, 384, 4), (5, 2, 384, 4), (5, 4, 384, 12), (5, 8, 384, 4), (7, 2, 512, 4), (5, 8, 96, 12), (5, 1, 384, 24), (5, 2, 384, 8)])
def test_fold_dwconv_ln_vs_numpy(eng, k, dil, C, S):
    rng = np.random.default_rng(100 * k + dil + C)
    seqlen = np.array([1, 5, 33, 64, 70, 150, 2, 31, 32, 96, 97], np.int32)
    M = int(seqlen.sum())
    x = rng.standard_normal((M, C)).astype(np.float32)
    part = bf16_round((0.5 * rng.standard_normal((S, M, C))).astype(np.float32))
    b2 = (0.3 * rng.standard_normal(C)).astype(np.float32)
    gamma = (0.5 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    rowvec = rng.standard_normal((len(seqlen), C)).astype(np.float32)
    w = (rng.standard_normal((C, k)) / np.sqrt(k)).astype(np.float32)
    bias = (0.1 * rng.standard_normal(C)).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    b = (0.1 * rng.standard_normal(C)).astype(np.float32)
    for rv in (rowvec, None):
        xo, y = eng.op_fold_dwconv_ln(seqlen, x, part, b2, gamma, rv, w, bias, g, b, k, dil)
        rxo, ry = _fold_dwconv_ref(seqlen, x, part, b2, gamma, rv, w, bias, g, b, k, dil)
        assert np.array_equal(xo, rxo)  # the fold is elementwise fp32 in a fixed order: exact
        mx, rms = rel_err(y, ry)
        assert rms < 4e-3 and mx < 4e-2, (mx, rms)  # one bf16 rounding of the normalised output (2^-9 of values up to ~5 sigma)


def test_fold_dwconv_ln_run_length_does_not_change_a_bit(eng):
    """Few sequences are cut into runs of 8 frames, many into runs of 32 (kernels_misc.hip, FOLD_TCH_FEW): a frame's arithmetic does not depend
    on the run it falls in, so three sequences alone and the same three among forty give the same bits."""
    rng = np.random.default_rng(77)
    C, S, k = 384, 12, 5
    lens_few = np.array([49, 70, 9], np.int32)
    lens_many = np.concatenate([lens_few, rng.integers(20, 90, 37).astype(np.int32)])
    assert len(lens_few) * ((lens_few.max() + 31) // 32) < 64 <= len(lens_many) * ((lens_many.max() + 31) // 32)
    M = int(lens_many.sum()); m = int(lens_few.sum())
    x = rng.standard_normal((M, C)).astype(np.float32)
    part = bf16_round((0.5 * rng.standard_normal((S, M, C))).astype(np.float32))
    b2 = (0.3 * rng.standard_normal(C)).astype(np.float32)
    gamma = (0.5 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    rowvec = rng.standard_normal((len(lens_many), C)).astype(np.float32)
    w = (rng.standard_normal((C, k)) / np.sqrt(k)).astype(np.float32)
    bias = (0.1 * rng.standard_normal(C)).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    b = (0.1 * rng.standard_normal(C)).astype(np.float32)
    for dil in (1, 8):
        xo1, y1 = eng.op_fold_dwconv_ln(lens_few, x[:m].copy(), np.ascontiguousarray(part[:, :m]), b2, gamma, rowvec[:3].copy(), w, bias, g, b, k, dil)
        xo2, y2 = eng.op_fold_dwconv_ln(lens_many, x, part, b2, gamma, rowvec, w, bias, g, b, k, dil)
        assert np.array_equal(xo1, xo2[:m]) and np.array_equal(y1, y2[:m]), dil


# ---- the fused forms INSIDE the engine: stage-ordered weight packing, FfnArgs wiring, fold state, every stage mask ----

def _mid_arch():
    """A small stack whose ConvNeXt widths are the ones K4 supports (384 / 512), so that stn_set_fused_ffn_min_rows(1, 1) puts every
    stage on the fused kernels with a handful of rows: text encoder 384/768 (mask 4), estimator 384/1536 (masks 2 and 8), vocoder
    512/1024 (mask 1)."""
    a = tiny_arch()
    a.te_dim, a.te_hidden, a.te_heads, a.te_ffn, a.te_conv_blocks, a.te_attn_blocks = 384, 768, 4, 256, 2, 1
    a.ve_dim, a.ve_hidden, a.ve_heads, a.ve_main_blocks, a.ve_dilated, a.ve_tail_blocks = 384, 1536, 4, 1, 3, 2
    a.vo_dim, a.vo_hidden, a.vo_blocks = 512, 1024, 2
    return a


@pytest.mark.parametrize("mode", ["bf16", "f16"])
def test_engine_stages_with_every_fused_form_on_and_off_against_the_oracle(mode):
    from oracle.neural_ref import RefModel, randn
    from gpu_util import make_inputs, parity_check
    a = _mid_arch()
    ref = RefModel(a, 5)
    ids, mask, sttl, sdp = make_inputs(a, 3, 14, [14, 9, 5], seed=8)
    durs = np.array([0.42, 0.30, 0.12], np.float32)
    nz = {}

    def nf(B, D, L):
        nz["x"] = randn(77, B, D, L)
        return nz["x"]

    rw, rd = ref.synthesize(ids, mask, sttl, sdp, 2, 1.05, nf, duration_override=durs)
    eng = binding.Engine(0, mode)
    eng.load_synthetic(a, 5)
    eng.set_fused_ffn_min_rows(1, 1)
    out = {}
    for packed in (True, False):
        eng.set_packed_rows(packed)
        for mask_bits in (0, 1, 2, 4, 7, 8, 15):
            eng.set_fused_ffn(mask_bits)
            w, d = eng.synthesize(ids, mask, sttl, sdp, 2, 1.05, noise=nz["x"], duration_override=durs)
            np.testing.assert_allclose(d, rd, rtol=1e-6)
            parity_check("ffn_forms.e2e_wav", mode, w, rw, "e2e")
            out[(packed, mask_bits)] = w
    # the forms differ from the two launches (and from each other) by rounding only, in both layouts
    for key, w in out.items():
        mx, rms = rel_err(w, out[(key[0], 0)])
        assert mx < (2e-1 if mode == "bf16" else 3e-2) and rms < (3e-2 if mode == "bf16" else 5e-3), (key, mx, rms)
    # K4-split exists in the packed layout only: in the padded one mask 8 changes nothing
    assert np.array_equal(out[(False, 8)], out[(False, 0)]) and not np.array_equal(out[(True, 8)], out[(True, 0)])
    assert not np.array_equal(out[(True, 2)], out[(True, 0)]) and not np.array_equal(out[(True, 4)], out[(True, 0)])
    eng.close()


def test_k4_split_graph_replay_is_bit_identical_to_eager():
    from gpu_util import make_inputs
    a = _mid_arch()
    eng = binding.Engine(0, "bf16")
    eng.load_synthetic(a, 5)
    eng.set_fused_ffn_min_rows(1, 1)
    eng.set_fused_ffn(15)
    ids, mask, sttl, sdp = make_inputs(a, 3, 14, [14, 9, 5], seed=8)
    durs = np.array([0.42, 0.30, 0.12], np.float32)
    eng.set_graph_mode(False)
    w0, _ = eng.synthesize(ids, mask, sttl, sdp, 3, 1.05, duration_override=durs, noise_seed=3)
    eng.set_graph_mode(True)
    r0 = eng.graph_replays
    for _ in range(4):
        w, _ = eng.synthesize(ids, mask, sttl, sdp, 3, 1.05, duration_override=durs, noise_seed=3)
        assert np.array_equal(w, w0)
    assert eng.graph_replays >= r0 + 2
    eng.close()
