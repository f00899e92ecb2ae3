"""The CPU oracle (oracle/stn_ref.c) against an independent statement of the four stage compositions in plain torch.nn.functional
(tools/torch_stages.py, written from the documented contract without reading the oracle's source; fixture
tests/golden/neural_tiny.json = its inputs and per-stage outputs on the tiny descriptor with the oracle's synthetic tensors).

This pins nothing to ONNX Runtime — the published graphs are unavailable offline, neural parity stays "unpinned"  —
but oracle and engine no longer share a single author's reading of mask placement, rotary pairing, residual order, the Euler
sign and the vocoder's un-compress mapping.  Call-site contract: /root/reference/cpp/helper.cpp:512-679, /root/reference/py/helper.py:177-215."""
import json
import os

import numpy as np
import pytest

from oracle.neural_ref import RefModel
from supertonic_amd.arch import tiny_arch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "neural_tiny.json")


def load_fixture():
    d = json.load(open(GOLD))
    dec = lambda e: np.asarray(e["data"], dtype=e["dtype"]).reshape(e["shape"])
    return {k: dec(v) for k, v in d["inputs"].items()}, {k: dec(v) for k, v in d["outputs"].items()}


def rel(a, b):
    rms = float(np.sqrt(np.mean(np.square(b.astype(np.float64))))) + 1e-30
    return float(np.max(np.abs(a.astype(np.float64) - b))) / rms


@pytest.fixture(scope="module")
def ref():
    return RefModel(tiny_arch(), 7)


def test_fixture_is_ragged_and_covers_the_edge_cases():
    inp, out = load_fixture()
    assert inp["text_mask"].sum(axis=(1, 2)).tolist() == [9, 5, 12] and inp["latent_mask"].sum(axis=(1, 2)).tolist() == [7, 4, 10]
    assert inp["text_ids"].max() >= tiny_arch().vocab_size  # an out-of-vocabulary id (zero row)
    assert set(out) == {"duration", "text_emb", "denoised", "wav"}


def test_duration_predictor_matches_the_torch_statement(ref):
    inp, out = load_fixture()
    got = ref.duration(inp["text_ids"], inp["style_dp"], inp["text_mask"])
    assert rel(got, out["duration"]) < 1e-5


def test_text_encoder_matches_the_torch_statement(ref):
    inp, out = load_fixture()
    got = ref.text_enc(inp["text_ids"], inp["style_ttl"], inp["text_mask"])
    assert rel(got, out["text_emb"]) < 1e-5
    # padding columns are exactly zero in both
    pad = inp["text_mask"][:, 0, :] < 0.5
    assert np.all(got.transpose(0, 2, 1)[pad] == 0) and np.all(out["text_emb"].transpose(0, 2, 1)[pad] == 0)


def test_vector_estimator_step_matches_the_torch_statement(ref):
    inp, out = load_fixture()
    got = ref.vector_est(inp["noisy"], out["text_emb"], inp["style_ttl"], inp["text_mask"], inp["latent_mask"], inp["total_step"],
                         inp["current_step"])
    assert rel(got, out["denoised"]) < 1e-5
    # and it is an Euler step with the velocity ADDED, scaled by 1 / total_step: the update is not zero and vanishes in the padding
    upd = got - inp["noisy"]
    assert np.abs(upd).max() > 1e-3 and np.all(got.transpose(0, 2, 1)[inp["latent_mask"][:, 0, :] < 0.5] == 0)


def test_vocoder_matches_the_torch_statement(ref):
    inp, out = load_fixture()
    got = ref.vocoder(out["denoised"])
    assert got.shape == out["wav"].shape and rel(got, out["wav"]) < 1e-5
