import numpy as np

from oracle import host_ref


def rel_err(got, ref):
    """(max |got-ref| / rms(ref), rms(got-ref) / rms(ref))"""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    rms = np.sqrt(np.mean(ref ** 2)) + 1e-30
    d = got - ref
    return float(np.abs(d).max() / rms), float(np.sqrt(np.mean(d ** 2)) / rms)


def make_inputs(a, B, Lt, lens, seed=0):
    rng = np.random.default_rng(seed)
    ids = rng.integers(1, a.vocab_size, (B, Lt)).astype(np.int64)
    mask = host_ref.length_to_mask(lens, Lt)
    ids = (ids * mask[:, 0, :]).astype(np.int64)
    sdp = (rng.standard_normal((B, a.n_style_dp, a.d_style_dp)) * 0.3).astype(np.float32)
    sttl = (rng.standard_normal((B, a.n_style_ttl, a.d_style_ttl)) * 0.3).astype(np.float32)
    return ids, mask, sttl, sdp
