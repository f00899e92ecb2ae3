#!/usr/bin/env python3
"""An independent second statement of the four stage compositions, as plain torch.nn.functional calls, and the generator of
tests/golden/neural_tiny.json.

Why: the CPU oracle (oracle/stn_ref.c) and the HIP engine were written from the same descriptor (include/stn_arch.h, DESIGN.md
section 3); only the oracle's primitives were checked against PyTorch.  This file restates the COMPOSITIONS — mask placement,
rotary pairing, residual order, Euler sign, the vocoder's un-compress mapping — from the documented contract (DESIGN.md section 3,
SURVEY.md Appendix A/C, the call sites /root/reference/cpp/helper.cpp:512-679 and /root/reference/py/helper.py:177-215) without
reading oracle/stn_ref.c, takes the oracle's synthetic tensors by name, and writes inputs + per-stage outputs of a tiny descriptor as
a fixture.  tests/test_neural_golden_cpu.py holds the oracle to it (1e-5) and tests/test_gpu_load_dir.py the engine.

It pins nothing to ONNX Runtime: the published graphs are not available offline ("parity unpinned").  What it
removes is the single-author blind spot between oracle and engine.

    python tools/torch_stages.py            # writes tests/golden/neural_tiny.json
"""
import json
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Weights:
    """Tensors by canonical name (Linear [N][K]; depthwise conv [C][k]; vocoder input conv [Cout][Cin][k]; vectors [n])."""

    def __init__(self, get):
        self._get = get

    def __call__(self, name, *shape):
        t = torch.from_numpy(np.asarray(self._get(name), np.float32).copy())
        return t.reshape(*shape) if shape else t


def length_mask(lens, L):
    """[B, L] float prefix mask."""
    return (torch.arange(L)[None, :] < torch.as_tensor(lens)[:, None]).float()


def convnext(W, p, x, m, C, hid, k, dil):
    """x [B, L, C]; m [B, L] (None: no mask).  x <- (x + gamma * pw2(GELU(pw1(LN(dwconv(x)))))) * mask."""
    h = F.conv1d(x.transpose(1, 2), W(p + ".dw.w", C, 1, k), W(p + ".dw.b"), padding=(k - 1) // 2 * dil, dilation=dil, groups=C).transpose(1, 2)
    h = F.layer_norm(h, (C,), W(p + ".ln.g"), W(p + ".ln.b"), eps=EPS)
    h = F.linear(F.gelu(F.linear(h, W(p + ".pw1.w", hid, C), W(p + ".pw1.b"))), W(p + ".pw2.w", C, hid), W(p + ".pw2.b"))
    x = x + W(p + ".gamma") * h
    return x if m is None else x * m[..., None]


def rope(t, pos, dh, base):
    """t [B, H, L, dh], pos [B, L] (float positions): rotate the pairs (i, i + dh/2) by pos * base^(-2i/dh)."""
    half = dh // 2
    inv = torch.exp(-math.log(base) * (2.0 * torch.arange(half).float()) / dh)
    ang = pos[:, None, :, None] * inv[None, None, None, :]
    c, s = torch.cos(ang), torch.sin(ang)
    a, b = t[..., :half], t[..., half:]
    return torch.cat([a * c - b * s, b * c + a * s], dim=-1)


def attention(W, p, x, qmask, ctx, kmask, C, H, Cctx, rope_mode, qlen=None, klen=None):
    """x [B, Lq, C] <- (x + Wo softmax(rope(q) rope(k)^T / sqrt(dh)) v) * qmask, q = Wq LN(x); k, v from ctx (self: ctx = LN(x)).
    rope_mode: -1 none, 0 position index, 1 length-aware (gamma * t / len, each side with its own length)."""
    B, Lq, _ = x.shape
    dh = C // H
    xn = F.layer_norm(x, (C,), W(p + ".ln.g"), W(p + ".ln.b"), eps=EPS)
    src = xn if ctx is None else ctx
    Lk = src.shape[1]
    q = F.linear(xn, W(p + ".q.w", C, C), W(p + ".q.b")).reshape(B, Lq, H, dh).transpose(1, 2)
    k = F.linear(src, W(p + ".k.w", C, Cctx), W(p + ".k.b")).reshape(B, Lk, H, dh).transpose(1, 2)
    v = F.linear(src, W(p + ".v.w", C, Cctx), W(p + ".v.b")).reshape(B, Lk, H, dh).transpose(1, 2)
    if rope_mode >= 0:
        pq = torch.arange(Lq).float()[None, :].expand(B, Lq)
        pk = torch.arange(Lk).float()[None, :].expand(B, Lk)
        if rope_mode == 1:
            pq = GAMMA * pq / torch.clamp(torch.as_tensor(qlen).float(), min=1)[:, None]
            pk = GAMMA * pk / torch.clamp(torch.as_tensor(klen).float(), min=1)[:, None]
        q, k = rope(q, pq, dh, ROPE_BASE), rope(k, pk, dh, ROPE_BASE)
    sc = q @ k.transpose(-1, -2) / math.sqrt(dh)
    if kmask is not None:
        sc = sc.masked_fill(kmask[:, None, None, :] < 0.5, float("-inf"))
    o = (torch.softmax(sc, dim=-1) @ v).transpose(1, 2).reshape(B, Lq, C)
    x = x + F.linear(o, W(p + ".o.w", C, C), W(p + ".o.b"))
    return x if qmask is None else x * qmask[..., None]


def embed(W, name, ids, m, vocab, C):
    emb = W(name, vocab, C)
    ok = ((ids >= 0) & (ids < vocab)).float() * m  # ids outside [0, vocab) give a zero row (the C++ host's policy)
    return emb[ids.clamp(0, vocab - 1)] * ok[..., None]


def duration(W, a, ids, style_dp, tmask):
    """duration_predictor: text_ids [B,Lt], style_dp [B,n,d], text_mask [B,1,Lt] -> seconds [B]  (cpp/helper.cpp:512-523)."""
    m = tmask[:, 0, :]
    C = a.dp_dim
    x = embed(W, "dp.emb", ids, m, a.vocab_size, C)
    for i in range(a.dp_conv_blocks):
        x = convnext(W, f"dp.conv{i}", x, m, C, a.dp_hidden, a.dp_kernel, 1)
    x = attention(W, "dp.st", x, m, style_dp, None, C, a.dp_heads, a.d_style_dp, -1)
    xn = F.layer_norm(x, (C,), W("dp.out_ln.g"), W("dp.out_ln.b"), eps=EPS)
    pooled = (xn * m[..., None]).sum(1) / torch.clamp(m.sum(1), min=1)[:, None]
    h = F.gelu(F.linear(pooled, W("dp.fc1.w", C, C), W("dp.fc1.b")))
    return F.softplus(F.linear(h, W("dp.fc2.w", 1, C), W("dp.fc2.b")))[:, 0]


def text_enc(W, a, ids, style_ttl, tmask):
    """text_encoder: -> text_emb [B, Ce, Lt]  (cpp/helper.cpp:545-556)."""
    m = tmask[:, 0, :]
    C = a.te_dim
    x = embed(W, "te.emb", ids, m, a.vocab_size, C)
    for i in range(a.te_conv_blocks):
        x = convnext(W, f"te.conv{i}", x, m, C, a.te_hidden, a.te_kernel, 1)
    for i in range(a.te_attn_blocks):
        p = f"te.sa{i}"
        x = attention(W, p, x, m, None, m, C, a.te_heads, C, 0)
        h = F.layer_norm(x, (C,), W(p + ".ffn_ln.g"), W(p + ".ffn_ln.b"), eps=EPS)
        h = F.linear(F.gelu(F.linear(h, W(p + ".ffn1.w", a.te_ffn, C), W(p + ".ffn1.b"))), W(p + ".ffn2.w", C, a.te_ffn), W(p + ".ffn2.b"))
        x = (x + h) * m[..., None]
    for i in range(a.te_style_blocks):
        x = attention(W, f"te.st{i}", x, m, style_ttl, None, C, a.te_heads, a.d_style_ttl, -1)
    xn = F.layer_norm(x, (C,), W("te.out_ln.g"), W("te.out_ln.b"), eps=EPS)
    out = F.linear(xn, W("te.proj.w", a.te_out_dim, C), W("te.proj.b")) * m[..., None]
    return out.transpose(1, 2)


def vector_est(W, a, noisy, text_emb, style_ttl, tmask, lmask, total_step, current_step):
    """vector_estimator, ONE Euler step inside the graph: -> denoised [B, D, L] = (noisy + v / total_step) * latent_mask
    (cpp/helper.cpp:620-658)."""
    C, H, nb = a.ve_dim, a.ve_heads, a.ve_main_blocks
    D = a.latent_dim * a.chunk_compress_factor
    m, tm = lmask[:, 0, :], tmask[:, 0, :]
    llen, tlen = m.sum(1), tm.sum(1)
    x = F.linear(noisy.transpose(1, 2), W("ve.in.w", C, D), W("ve.in.b")) * m[..., None]
    ctx = text_emb.transpose(1, 2)  # [B, Lt, Ce]
    # time conditioning: sinusoid(t * scale) -> Linear -> SiLU -> Linear, then one Linear per main block
    t = current_step / total_step * TIME_SCALE
    hd = a.ve_time_dim // 2
    f = torch.exp(-math.log(10000.0) * torch.arange(hd).float() / hd)
    te = torch.cat([torch.sin(t[:, None] * f[None, :]), torch.cos(t[:, None] * f[None, :])], dim=1)
    tc = F.linear(F.silu(F.linear(te, W("ve.t1.w", C, a.ve_time_dim), W("ve.t1.b"))), W("ve.t2.w", C, C), W("ve.t2.b"))
    for b in range(nb):
        p = f"ve.m{b}"
        for j in range(a.ve_dilated):
            x = convnext(W, f"{p}.dil{j}", x, m, C, a.ve_hidden, a.ve_kernel, 1 << j)
        x = (x + F.linear(tc, W(p + ".time.w", C, C), W(p + ".time.b"))[:, None, :]) * m[..., None]
        x = convnext(W, p + ".cn_a", x, m, C, a.ve_hidden, a.ve_kernel, 1)
        x = attention(W, p + ".text", x, m, ctx, tm, C, H, a.te_out_dim, 1, llen, tlen)
        x = convnext(W, p + ".cn_b", x, m, C, a.ve_hidden, a.ve_kernel, 1)
        x = attention(W, p + ".style", x, m, style_ttl, None, C, H, a.d_style_ttl, -1)
    for j in range(a.ve_tail_blocks):
        x = convnext(W, f"ve.tail{j}", x, m, C, a.ve_hidden, a.ve_kernel, 1)
    xn = F.layer_norm(x, (C,), W("ve.out_ln.g"), W("ve.out_ln.b"), eps=EPS)
    v = F.linear(xn, W("ve.out.w", D, C), W("ve.out.b")).transpose(1, 2)  # [B, D, L]
    return (noisy + v / total_step[:, None, None]) * lmask


def vocoder(W, a, latent):
    """vocoder: latent [B, D = ld*ccf, L] -> wav [B, L*ccf*hop]; no mask input: the batch is decoded densely (cpp/helper.cpp:662-672).
    Frame t = l*ccf + q of utterance b has channel c = latent[b][q*ld + c][l]."""
    B, D, L = latent.shape
    ld, ccf, C = a.latent_dim, a.chunk_compress_factor, a.vo_dim
    fr = latent.reshape(B, ccf, ld, L).permute(0, 3, 1, 2).reshape(B, L * ccf, ld)  # [B, T, ld]
    x = F.conv1d(fr.transpose(1, 2), W("vo.in.w", C, ld, a.vo_in_kernel), W("vo.in.b"), padding=(a.vo_in_kernel - 1) // 2).transpose(1, 2)
    for i in range(a.vo_blocks):
        x = convnext(W, f"vo.blk{i}", x, None, C, a.vo_hidden, a.vo_kernel, int(a.vo_dilations[i]))
    xn = F.layer_norm(x, (C,), W("vo.out_ln.g"), W("vo.out_ln.b"), eps=EPS)
    return F.linear(xn, W("vo.head.w", a.base_chunk_size, C), W("vo.head.b")).reshape(B, -1)


def configure(a):
    global EPS, ROPE_BASE, GAMMA, TIME_SCALE
    EPS, ROPE_BASE, GAMMA, TIME_SCALE = float(a.ln_eps), float(a.rope_base), float(a.larope_gamma), float(a.time_scale)


def fixture_inputs(a, seed=2024):
    """Small ragged batch: B = 3, text lengths 9 / 5 / 12, latent lengths 7 / 4 / 10 (an id outside the vocabulary included)."""
    rng = np.random.default_rng(seed)
    B, Lt, L = 3, 12, 10
    tl, ll = np.array([9, 5, 12]), np.array([7, 4, 10])
    tmask = (np.arange(Lt)[None, :] < tl[:, None]).astype(np.float32)[:, None, :]
    lmask = (np.arange(L)[None, :] < ll[:, None]).astype(np.float32)[:, None, :]
    ids = (rng.integers(1, a.vocab_size, (B, Lt)) * tmask[:, 0, :]).astype(np.int64)
    ids[0, 3] = a.vocab_size + 5  # out of range -> zero row
    D = a.latent_dim * a.chunk_compress_factor
    return dict(text_ids=ids, text_mask=tmask, latent_mask=lmask,
                style_dp=(0.3 * rng.standard_normal((B, a.n_style_dp, a.d_style_dp))).astype(np.float32),
                style_ttl=(0.3 * rng.standard_normal((B, a.n_style_ttl, a.d_style_ttl))).astype(np.float32),
                noisy=(rng.standard_normal((B, D, L)).astype(np.float32) * lmask),
                total_step=np.full(B, 4, np.float32), current_step=np.array([0, 1, 3], np.float32))


def run_all(W, a, inp):
    T = lambda k: torch.from_numpy(inp[k])
    configure(a)
    with torch.no_grad():
        dur = duration(W, a, T("text_ids"), T("style_dp"), T("text_mask"))
        emb = text_enc(W, a, T("text_ids"), T("style_ttl"), T("text_mask"))
        den = vector_est(W, a, T("noisy"), emb, T("style_ttl"), T("text_mask"), T("latent_mask"), T("total_step"), T("current_step"))
        wav = vocoder(W, a, den)
    return dict(duration=dur.numpy(), text_emb=emb.numpy(), denoised=den.numpy(), wav=wav.numpy())


def main():
    from oracle.neural_ref import RefModel  # (only its synthetic TENSORS are taken, by name)
    from supertonic_amd.arch import tiny_arch
    torch.set_num_threads(4)
    a = tiny_arch()
    ref = RefModel(a, 7)
    inp = fixture_inputs(a)
    out = run_all(Weights(ref.tensor), a, inp)
    enc = lambda v: {"shape": list(v.shape), "dtype": str(v.dtype), "data": [float(x) if v.dtype != np.int64 else int(x) for x in v.ravel()]}
    doc = {"_comment": "inputs and per-stage outputs of tools/torch_stages.py (torch %s, CPU fp32) on the tiny descriptor with the oracle's synthetic "
                       "tensors (seed 7); generated data, regenerate with `python tools/torch_stages.py`" % torch.__version__,
           "arch": "tiny_arch()", "weight_seed": 7, "inputs": {k: enc(v) for k, v in inp.items()}, "outputs": {k: enc(v) for k, v in out.items()}}
    path = os.path.join(ROOT, "tests", "golden", "neural_tiny.json")
    with open(path, "w") as f:
        json.dump(doc, f)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
