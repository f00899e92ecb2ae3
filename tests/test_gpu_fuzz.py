"""Randomised differential test: engine vs CPU oracle over random small shapes (batch, text lengths, durations, Euler steps,
speed, vocoder mode, injected or device noise), tiny architecture, all three arithmetic modes.  Seeds are fixed: the sweep is
deterministic.  STN_FUZZ_CASES=<n> lengthens it (the round-1 soak ran 2000 cases per mode: worst max error 3.4e-6 fp32,
3.6e-2 bf16, 4.8e-3 f16).  The duration predictor runs in exact fp32 in every mode (round 2), so predicted durations — and with
them L, the latent lengths and the trim points — are the fp32 oracle's in 16-bit engines too: no frame-boundary flips."""
import os

import numpy as np
import pytest

from oracle import host_ref
from oracle.neural_ref import RefModel, randn
from supertonic_amd import binding
from supertonic_amd.arch import tiny_arch
from gpu_util import make_inputs, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,tol_max,tol_rms", [("f32", 1e-4, 3e-5), ("bf16", 1.5e-1, 4e-2), ("f16", 2e-2, 5e-3)])
def test_random_shapes_against_oracle(dtype, tol_max, tol_rms):
    a = tiny_arch()
    ref = RefModel(a, 7)
    eng = binding.Engine(0, dtype)
    eng.load_synthetic(a, 7)
    rng = np.random.default_rng(20260001)
    cs = a.base_chunk_size * a.chunk_compress_factor
    worst = (0.0, None)
    flips = 0  # predicted-duration cases whose latent length flipped by one frame in 16-bit arithmetic
    for case in range(int(os.environ.get("STN_FUZZ_CASES", "24"))):
        B = int(rng.integers(1, 6))
        Lt = int(rng.integers(1, 40))
        lens = rng.integers(1, Lt + 1, B)
        lens[rng.integers(0, B)] = Lt
        if case % 7 == 3:
            lens[0] = 0  # an empty text in the batch (cpp/helper.cpp:366-376 allows it)
            if B == 1:
                lens[0] = Lt
        ids, mask, sttl, sdp = make_inputs(a, B, Lt, lens, seed=100 + case)
        steps = int(rng.integers(1, 5))
        speed = float(rng.choice([0.8, 1.0, 1.05, 1.5]))
        use_pred = case % 5 == 4  # predicted durations instead of forced ones
        durs = None if use_pred else rng.uniform(0.05, 1.2, B).astype(np.float32)
        nz = {}

        def nf(Bn, D, L):
            nz["x"] = randn(1000 + case, Bn, D, L)
            return nz["x"]

        rw, rd = ref.synthesize(ids, mask, sttl, sdp, steps, speed, nf, duration_override=durs)
        inject = case % 2 == 0
        eng.set_vocoder_mode(False)
        if inject:
            try:
                w, d = eng.synthesize(ids, mask, sttl, sdp, steps, speed, noise=nz["x"], duration_override=durs)
            except binding.StnError as err:
                # 16-bit predicted durations put the longest utterance one frame across a boundary: the oracle's noise tensor
                # no longer has the engine's L (the engine refuses it by design)
                if use_pred and dtype != "f32" and "injected noise has L=" in str(err):
                    flips += 1
                    continue
                raise
        else:  # device Philox with the oracle's (seed, utterance) counters
            w, d = eng.synthesize(ids, mask, sttl, sdp, steps, speed, duration_override=durs, noise_seed=1000 + case)
        assert w.shape == rw.shape, (case, w.shape, rw.shape)
        np.testing.assert_allclose(d, rd, rtol=1e-5, err_msg=str(case))  # fp32 predictor in every mode
        if use_pred and dtype != "f32":
            # durations predicted in 16-bit arithmetic can land on the other side of a latent-frame boundary
            # (len = ceil(floor(dur * sr) / chunk)): such an utterance legitimately has one frame more or less than the oracle's
            le = host_ref.latent_geometry(np.asarray(d, np.float32), a.sample_rate, a.base_chunk_size, a.chunk_compress_factor, a.latent_dim)[2]
            lr_ = host_ref.latent_geometry(np.asarray(rd, np.float32), a.sample_rate, a.base_chunk_size, a.chunk_compress_factor, a.latent_dim)[2]
            if not np.array_equal(le, lr_):
                assert np.all(np.isfinite(w)) and np.abs(np.asarray(le) - np.asarray(lr_)).max() <= 1, (case, le, lr_)
                flips += 1
                continue
        mx, rms = rel_err(w, rw)
        assert np.all(np.isfinite(w)) and mx < tol_max and rms < tol_rms, (case, B, Lt, lens, steps, speed, mx, rms)
        if mx > worst[0]:
            worst = (mx, case)
        # length-aware mode on the same inputs: every row's own frames equal the padded-batch result wherever the receptive
        # field does not reach the padding, and everything past the row's length is exactly zero
        if case % 3 == 0 and not use_pred:
            eng.set_vocoder_mode(True)
            w2, d2 = (eng.synthesize(ids, mask, sttl, sdp, steps, speed, noise=nz["x"], duration_override=durs) if inject else
                      eng.synthesize(ids, mask, sttl, sdp, steps, speed, duration_override=durs, noise_seed=1000 + case))
            eng.set_vocoder_mode(False)
            _, _, ll = host_ref.latent_geometry(np.asarray(d2, np.float32), a.sample_rate, a.base_chunk_size, a.chunk_compress_factor, a.latent_dim)
            for b in range(B):
                assert np.all(w2[b, int(ll[b]) * cs:] == 0.0), (case, b)
    print("worst case", worst, "frame-boundary flips", flips)
    # a flip now needs dur * sr within ~1e-6 (relative) of a chunk boundary: none in the fixed sweep, vanishingly few in a soak
    assert flips <= int(0.002 * int(os.environ.get("STN_FUZZ_CASES", "24")))
