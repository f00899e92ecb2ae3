"""The head-split cross-attention block of the vector estimator (kernels_xattn_hs.hip: per (utterance pair, head) the q projection, its
rotation, the attention and the head's share of the output projection in one launch, four 16-bit per-head partial sums folded by the next
ConvNeXt block) against the four-launch form it replaces, and against the CPU oracle.  Both forms round the same intermediates to
16 bits (LN output, q, rotated q, probabilities, attention output)."""
import numpy as np
import pytest

from oracle.neural_ref import RefModel, randn
from supertonic_amd import binding, host, workload
from supertonic_amd.arch import default_arch
from gpu_util import rel_err

pytestmark = pytest.mark.gpu


def _inputs(n, min_words, max_words, seed):
    arch = default_arch()
    texts = workload.utterances(n, min_words=min_words, max_words=max_words, seed=seed)
    ids, mask = host.UnicodeProcessor(host.synthetic_indexer())(texts, ["en"] * n)
    sttl, sdp = workload.synthetic_styles(arch, list(range(n)))
    durs = workload.forced_durations(texts) / np.float32(1.05)
    D, L, lens = host.latent_geometry(durs, 44100, 512, 6, 24)
    lm = (np.arange(L)[None, None, :] < np.asarray(lens)[:, None, None]).astype(np.float32)
    return arch, ids, mask, sttl, sdp, durs, D, L, lens, lm


# ---- the head-split form (kernels_xattn_hs.hip, the default): fold_ln + ONE launch per block, the output projection as four
# 16-bit per-head partial sums that the next ConvNeXt block's fold adds to x --------------------------------------------------------------
def _batch_latent(eng, mode, ids, mask, sttl, sdp, fd, steps, noise):
    eng.set_fused_xattn(mode)
    eng.batch_upload(ids, mask, sttl, sdp, duration_override=fd)
    eng.batch_set_noise(noise)
    eng.batch_run(steps, 1.05, 1234)
    return eng.batch_fetch_latent().copy()


@pytest.mark.parametrize("dtype,tol_max,tol_rms", [("bf16", 6e-2, 8e-3), ("f16", 8e-3, 1e-3)])
@pytest.mark.parametrize("n,min_words,max_words,dur_scale", [(5, 1, 9, 1.0), (70, 4, 12, 1.0), (67, 2, 13, 1.0), (66, 10, 15, 1.0), (3, 14, 16, 2.0)])
def test_head_split_equals_four_launches(dtype, tol_max, tol_rms, n, min_words, max_words, dur_scale):
    """Packed resident batch, two Euler steps: the latent of the head-split blocks against the four-launch blocks.  The two forms round
    the same intermediates (q, rotated q, exponentials, attention output) and differ by the 16-bit rounding of the per-head partial sums.
    Cases: one utterance per workgroup (5 and 3 utterances; the 3 long ones have up to 6 row tiles: two per wave), pairs of utterances
    per workgroup chosen longest-with-shortest, their rows one virtual sequence (70 short ones; 67: an odd count, the median utterance
    alone in its workgroup, and very short ones — a pair can fit one tile), pairs with more than four tiles (66 utterances of up to 103
    frames: some waves own two tiles, most tiles straddle the two utterances)."""
    arch, ids, mask, sttl, sdp, durs, D, L, lens, lm = _inputs(n, min_words, max_words, 100 + n)
    if dur_scale != 1.0:  # slower speech: more latent frames for the same text (more than four row tiles per utterance)
        durs = durs * np.float32(dur_scale)
        D, L, lens = host.latent_geometry(durs, 44100, 512, 6, 24)
        lm = (np.arange(L)[None, None, :] < np.asarray(lens)[:, None, None]).astype(np.float32)
    assert L <= 256 and ids.shape[1] <= 128, (L, ids.shape)  # inside the head-split kernel's shapes (the fallback has its own test)
    fd = durs * np.float32(1.05)
    noise = randn(17, n, D, L)
    eng = binding.Engine(0, dtype)
    eng.load_synthetic(arch, 7)
    lat = {m: _batch_latent(eng, m, ids, mask, sttl, sdp, fd, 2, noise) for m in (0, 1)}
    assert np.all(np.isfinite(lat[1]))
    mx, rms = rel_err(lat[1], lat[0])
    print(f"head-split vs four launches [{dtype}, n={n}, L={L}]: max {mx:.3e} rms {rms:.3e}")
    assert mx < tol_max and rms < tol_rms, (mx, rms)
    assert np.all(lat[1][lm.repeat(D, axis=1) == 0] == 0)  # padding frames stay exactly zero
    for _ in range(3):  # eager, captured, replayed: the same bits
        assert np.array_equal(_batch_latent(eng, 1, ids, mask, sttl, sdp, fd, 2, noise), lat[1])
    eng.set_fused_xattn(0)


@pytest.mark.parametrize("dtype,tol_max,tol_rms", [("bf16", 3e-1, 5e-2), ("f16", 4e-2, 8e-3)])
def test_head_split_vs_oracle(dtype, tol_max, tol_rms):
    arch, ids, mask, sttl, sdp, durs, D, L, lens, lm = _inputs(6, 3, 9, 5)
    ref = RefModel(arch, 7)
    nz = {}

    def nf(B, Dn, Ln):
        nz["x"] = randn(11, B, Dn, Ln)
        return nz["x"]

    fd = durs * np.float32(1.05)
    ref_wav, _ = ref.synthesize(ids, mask, sttl, sdp, 3, 1.05, nf, duration_override=fd)
    eng = binding.Engine(0, dtype)
    eng.load_synthetic(arch, 7)
    eng.set_fused_xattn(1)
    wav, _ = eng.synthesize(ids, mask, sttl, sdp, 3, 1.05, noise=nz["x"], duration_override=fd)
    mx, rms = rel_err(wav, ref_wav)
    print(f"head-split vs oracle [{dtype}]: max {mx:.3e} rms {rms:.3e}")
    assert mx < tol_max and rms < tol_rms, (mx, rms)
    eng.set_fused_xattn(0)


def test_head_split_falls_back_outside_its_shapes():
    """A text of more than 128 tokens (several key chunks) takes the four launches for the text blocks and the head-split kernel for the
    style blocks; the padded row layout takes the four launches everywhere: same results as mode 0 to rounding / exactly."""
    arch, ids, mask, sttl, sdp, durs, D, L, lens, lm = _inputs(4, 30, 40, 9)
    assert ids.shape[1] > 128
    fd = durs * np.float32(1.05)
    noise = randn(5, 4, D, L)
    eng = binding.Engine(0, "bf16")
    eng.load_synthetic(arch, 7)
    a = _batch_latent(eng, 0, ids, mask, sttl, sdp, fd, 2, noise)
    b = _batch_latent(eng, 1, ids, mask, sttl, sdp, fd, 2, noise)
    mx, rms = rel_err(b, a)
    assert np.all(np.isfinite(b)) and mx < 6e-2 and rms < 8e-3, (mx, rms)
    eng.set_packed_rows(False)
    a = _batch_latent(eng, 0, ids, mask, sttl, sdp, fd, 2, noise)
    b = _batch_latent(eng, 1, ids, mask, sttl, sdp, fd, 2, noise)
    assert np.array_equal(a, b)
    eng.set_packed_rows(True)
    eng.set_fused_xattn(0)
