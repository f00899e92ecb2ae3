"""The parity bounds the GPU tests assert (tests/golden/parity_bounds.json) against the errors measured on an MI355X
(profiles/parity_r04.json, written by tools/parity_record.py): every bound is at most 2x its measurement (or the fp32
summation-noise floor, max 2e-6 / rms 5e-7 of the output's rms) and at least the measurement itself, every measured case has a bound, and no bound is looser than the arithmetic mode's ceiling."""
import json
import os

from gpu_util import BOUNDS_PATH, CEILING

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bounds_are_at_most_twice_the_measured_error():
    with open(os.path.join(ROOT, "profiles", "parity_r04.json")) as f:
        measured = json.load(f)["measured"]
    with open(BOUNDS_PATH) as f:
        bounds = json.load(f)["bounds"]
    assert set(bounds) == set(measured) and len(measured) >= 12
    seen = set()
    for case, per in measured.items():
        assert set(bounds[case]) == set(per), case
        for dtype, m in per.items():
            seen.add(dtype)
            b = bounds[case][dtype]
            for k in ("max", "rms"):
                assert m[k] <= b[k] <= max(2.0 * m[k] * (1 + 1e-12), {"max": 2e-6, "rms": 5e-7}[k]), (case, dtype, k, m[k], b[k])
            cmax, crms = CEILING[dtype][m["kind"]]
            assert m["max"] <= cmax and m["rms"] <= crms, (case, dtype)  # the ceiling itself holds for what was measured
    assert seen == {"f32", "bf16", "f16"}
    assert {"c3.full_size_wav", "tiny.e2e_wav", "c1.full_single_utterance_wav"} <= set(measured)
