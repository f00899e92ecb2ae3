"""oracle/stn_ref.c primitives against plain PyTorch fp32 CPU definitions of the same ops,
plus structural properties of the four stages (shapes, masking, batch independence).

The neural oracle is "parity unpinned" w.r.t. the reference's ONNX graphs (absent); these
tests pin it to the textbook definitions of the ops it is built from."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import host_ref
from oracle.neural_ref import RefModel, lib, randn
from supertonic_amd.arch import default_arch, tiny_arch

torch.set_num_threads(4)


@pytest.fixture(scope="module")
def tiny():
    return RefModel(tiny_arch(), 7)


def test_param_count_default():
    m = RefModel(default_arch(), 7)
    # README.md:60 of the reference: 66 M parameters
    assert abs(m.param_count - 66e6) / 66e6 < 0.10, m.param_count


def test_weights_deterministic(tiny):
    other = RefModel(tiny_arch(), 7)
    assert np.array_equal(tiny.tensor("vo.blk0.pw1.w"), other.tensor("vo.blk0.pw1.w"))
    third = RefModel(tiny_arch(), 8)
    assert not np.array_equal(tiny.tensor("vo.blk0.pw1.w"), third.tensor("vo.blk0.pw1.w"))
    w = tiny.tensor("ve.m0.dil0.pw1.w")
    assert abs(w.std() - (1.0 / np.sqrt(96))) < 0.01  # unit-gain fan-in init
    g = tiny.tensor("ve.m0.dil0.ln.g")
    assert 0.85 < g.min() and g.max() < 1.15


def test_linear_vs_torch():
    rng = np.random.default_rng(1)
    X = rng.standard_normal((37, 48)).astype(np.float32)
    W = rng.standard_normal((20, 48)).astype(np.float32)
    b = rng.standard_normal(20).astype(np.float32)
    Y = np.empty((37, 20), np.float32)
    lib().stnref_linear(X, 37, 48, np.ascontiguousarray(W.T), b.ctypes.data, 20, Y)
    ref = torch.nn.functional.linear(torch.from_numpy(X), torch.from_numpy(W), torch.from_numpy(b)).numpy()
    np.testing.assert_allclose(Y, ref, rtol=1e-5, atol=1e-5)


def test_layernorm_vs_torch():
    rng = np.random.default_rng(2)
    X = (rng.standard_normal((19, 96)) * 3 + 1).astype(np.float32)
    g = rng.standard_normal(96).astype(np.float32)
    b = rng.standard_normal(96).astype(np.float32)
    Y = np.empty_like(X)
    lib().stnref_layernorm(X, 19, 96, g, b, 1e-6, Y)
    ref = torch.nn.functional.layer_norm(torch.from_numpy(X), (96,), torch.from_numpy(g), torch.from_numpy(b), 1e-6).numpy()
    np.testing.assert_allclose(Y, ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("k,dil", [(5, 1), (5, 8), (7, 4), (7, 1)])
def test_dwconv_vs_torch(k, dil):
    rng = np.random.default_rng(3)
    B, L, C = 3, 23, 16
    X = rng.standard_normal((B, L, C)).astype(np.float32)
    w = rng.standard_normal((C, k)).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    Y = np.empty_like(X)
    lib().stnref_dwconv(X.reshape(B * L, C), B, L, C, w, b, k, dil, Y.reshape(B * L, C))
    xt = torch.from_numpy(X).permute(0, 2, 1)  # NCL
    ref = torch.nn.functional.conv1d(xt, torch.from_numpy(w).unsqueeze(1), torch.from_numpy(b),
                                     padding=dil * (k - 1) // 2, dilation=dil, groups=C).permute(0, 2, 1).numpy()
    np.testing.assert_allclose(Y, ref, rtol=1e-5, atol=1e-5)


def test_attention_core_vs_torch():
    rng = np.random.default_rng(4)
    B, Lq, Lk, C, H = 2, 7, 11, 32, 4
    Q = rng.standard_normal((B, Lq, C)).astype(np.float32)
    K = rng.standard_normal((B, Lk, C)).astype(np.float32)
    V = rng.standard_normal((B, Lk, C)).astype(np.float32)
    klen = np.array([11, 6], np.int32)
    O = np.empty_like(Q)
    lib().stnref_attention_core(Q.reshape(-1, C), K.reshape(-1, C), V.reshape(-1, C), B, Lq, Lk, C, H,
                                klen.ctypes.data, O.reshape(-1, C))
    q = torch.from_numpy(Q).view(B, Lq, H, C // H).transpose(1, 2)
    k = torch.from_numpy(K).view(B, Lk, H, C // H).transpose(1, 2)
    v = torch.from_numpy(V).view(B, Lk, H, C // H).transpose(1, 2)
    mask = (torch.arange(Lk)[None, :] < torch.from_numpy(klen)[:, None])[:, None, None, :]
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=mask).transpose(1, 2).reshape(B, Lq, C).numpy()
    np.testing.assert_allclose(O, ref, rtol=1e-4, atol=1e-5)


def test_randn_statistics_and_locality():
    n = randn(1234, 3, 144, 200)
    assert abs(n.mean()) < 0.01 and abs(n.std() - 1) < 0.01
    # element (utt, d, t) depends only on (seed, utt, d, t): sharding-invariant noise
    n2 = randn(1234, 1, 144, 77, utt_ids=[2])
    assert np.array_equal(n2[0], n[2, :, :77])
    assert not np.array_equal(randn(1235, 1, 144, 77)[0], n[0, :, :77])


def _inputs(a, B, Lt, lens, seed=0):
    rng = np.random.default_rng(seed)
    ids = rng.integers(1, a.vocab_size, (B, Lt)).astype(np.int64)
    mask = host_ref.length_to_mask(lens, Lt)
    ids = (ids * mask[:, 0, :]).astype(np.int64)
    sdp = (rng.standard_normal((B, a.n_style_dp, a.d_style_dp)) * 0.3).astype(np.float32)
    sttl = (rng.standard_normal((B, a.n_style_ttl, a.d_style_ttl)) * 0.3).astype(np.float32)
    return ids, mask, sttl, sdp


def test_stage_shapes_and_masking(tiny):
    a = tiny.arch
    ids, mask, sttl, sdp = _inputs(a, 3, 14, [14, 9, 5])
    dur = tiny.duration(ids, sdp, mask)
    assert dur.shape == (3,) and np.all(dur > 0) and np.all(np.isfinite(dur))
    emb = tiny.text_enc(ids, sttl, mask)
    assert emb.shape == (3, a.te_out_dim, 14)
    assert np.all(emb[1, :, 9:] == 0) and np.all(emb[2, :, 5:] == 0) and np.abs(emb[1, :, :9]).min() > 0
    L = 8
    lmask = host_ref.length_to_mask([8, 5, 3], L)
    x = randn(5, 3, a.latent_channels, L) * lmask
    ts, cs = np.full(3, 4, np.float32), np.full(3, 1, np.float32)
    y = tiny.vector_est(x, emb, sttl, mask, lmask, ts, cs)
    assert y.shape == x.shape and np.all(y[1, :, 5:] == 0) and np.all(np.isfinite(y))
    wav = tiny.vocoder(y)
    assert wav.shape == (3, L * a.chunk_size) and np.all(np.isfinite(wav))


def test_padding_does_not_change_valid_outputs(tiny):
    """Masked stages: an utterance's outputs must not depend on padding or on batch mates."""
    a = tiny.arch
    ids, mask, sttl, sdp = _inputs(a, 2, 12, [12, 7], seed=3)
    emb = tiny.text_enc(ids, sttl, mask)
    dur = tiny.duration(ids, sdp, mask)
    # utterance 1 alone, unpadded
    emb1 = tiny.text_enc(ids[1:, :7], sttl[1:], mask[1:, :, :7])
    dur1 = tiny.duration(ids[1:, :7], sdp[1:], mask[1:, :, :7])
    np.testing.assert_allclose(emb[1, :, :7], emb1[0], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dur[1], dur1[0], rtol=1e-5)
    L = 9
    lmask = host_ref.length_to_mask([9, 4], L)
    x = randn(9, 2, a.latent_channels, L) * lmask
    ts, cs = np.full(2, 5, np.float32), np.full(2, 2, np.float32)
    y = tiny.vector_est(x, emb, sttl, mask, lmask, ts, cs)
    y1 = tiny.vector_est(x[1:, :, :4], emb1, sttl[1:], mask[1:, :, :7], lmask[1:, :, :4], ts[1:], cs[1:])
    np.testing.assert_allclose(y[1, :, :4], y1[0], rtol=1e-4, atol=1e-5)


def test_euler_step_convention(tiny):
    """denoised = (x + v/total_step) * mask: v is independent of total_step's scale only through t."""
    a = tiny.arch
    ids, mask, sttl, sdp = _inputs(a, 1, 10, [10], seed=5)
    emb = tiny.text_enc(ids, sttl, mask)
    L = 6
    lmask = host_ref.length_to_mask([6], L)
    x = randn(3, 1, a.latent_channels, L)
    # same t = 0 (current_step 0) with total_step 2 vs 4 -> same v, different dt
    y2 = tiny.vector_est(x, emb, sttl, mask, lmask, np.array([2.], np.float32), np.array([0.], np.float32))
    y4 = tiny.vector_est(x, emb, sttl, mask, lmask, np.array([4.], np.float32), np.array([0.], np.float32))
    np.testing.assert_allclose((y2 - x) * 2, (y4 - x) * 4, rtol=1e-3, atol=1e-5)


def test_synthesize_end_to_end(tiny):
    a = tiny.arch
    ids, mask, sttl, sdp = _inputs(a, 2, 12, [12, 8], seed=7)
    wav, dur = tiny.synthesize(ids, mask, sttl, sdp, 3, 1.05, lambda B, D, L: randn(1234, B, D, L))
    D, L, lat = host_ref.latent_geometry(dur, a.sample_rate, a.base_chunk_size, a.chunk_compress_factor, a.latent_dim)
    assert wav.shape == (2, L * a.chunk_size) and np.all(np.isfinite(wav)) and 0.01 < wav.std() < 1.0
