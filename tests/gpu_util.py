import json
import os

import numpy as np

from oracle import host_ref

_HERE = os.path.dirname(os.path.abspath(__file__))
BOUNDS_PATH = os.path.join(_HERE, "golden", "parity_bounds.json")
# Generic ceilings (relative to the oracle output's rms) — what the arithmetic mode can promise; a case listed in
# tests/golden/parity_bounds.json is held to ITS bound instead: <= 2x what tools/parity_record.py measured on an MI355X
# (profiles/parity_r04.json), never looser than the ceiling.
CEILING = {"f32": dict(stage=(2e-4, 5e-5), e2e=(2e-3, 5e-4)),
           "bf16": dict(stage=(1e-1, 2e-2), e2e=(3e-1, 5e-2)),
           "f16": dict(stage=(1.5e-2, 3e-3), e2e=(4e-2, 8e-3))}


def _bounds():
    if not hasattr(_bounds, "v"):
        try:
            with open(BOUNDS_PATH) as f:
                _bounds.v = json.load(f)["bounds"]
        except FileNotFoundError:
            _bounds.v = {}
    return _bounds.v


def parity_check(case, dtype, got, ref, kind="stage"):
    """Engine output against the oracle's for one named case: finite, and (max, rms) error within the case's bound.  With
    STN_PARITY_RECORD=<file> every measurement is appended there as a JSON line (tools/parity_record.py)."""
    mx, rms = rel_err(got, ref)
    assert np.all(np.isfinite(np.asarray(got))), (case, dtype)
    rec = os.environ.get("STN_PARITY_RECORD")
    if rec:
        with open(rec, "a") as f:
            f.write(json.dumps({"case": case, "dtype": dtype, "kind": kind, "max": mx, "rms": rms, "n": int(np.asarray(ref).size)}) + "\n")
    cmax, crms = CEILING[dtype][kind]
    b = _bounds().get(case, {}).get(dtype)
    bmax, brms = (min(b["max"], cmax), min(b["rms"], crms)) if b else (cmax, crms)
    if os.environ.get("STN_PARITY_RECORD_ONLY") == "1":  # measuring: only the ceiling applies
        bmax, brms = cmax, crms
    assert mx <= bmax and rms <= brms, f"{case} [{dtype}]: max {mx:.3e} (bound {bmax:.3g}) rms {rms:.3e} (bound {brms:.3g})"
    return mx, rms


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
