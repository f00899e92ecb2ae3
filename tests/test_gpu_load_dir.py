"""stn_load_dir end to end on a real MI355X: an asset directory with the reference's file set (cpp/helper.cpp:784-823) is
synthesised from the oracle's weights — four hand-encoded .onnx files (initializers in raw / packed / transposed-MatMul /
fp64 storage), tts.json, unicode_indexer.json and the weight manifest — and loaded through the built-in protobuf reader.
The engine loaded from files must reproduce the engine with the same weights generated in place, bit for bit (fp32)."""
import json

import numpy as np
import pytest

from oracle.neural_ref import RefModel, randn
from supertonic_amd import binding, host
from supertonic_amd.arch import tiny_arch
from gpu_util import make_inputs
import onnx_writer as ow

pytestmark = pytest.mark.gpu

FILES = {"dp": "duration_predictor.onnx", "te": "text_encoder.onnx", "ve": "vector_estimator.onnx", "vo": "vocoder.onnx"}


def _shape(name, a, n):
    """2-D view used for the transposed-storage variant (Linear weights only)."""
    return None


def build_asset_dir(tmp, a, ref, eng):
    names = eng.tensor_names(a)
    assert set(names) == set(ref.tensor_names())  # engine and oracle agree on the canonical tensor set
    per_file = {k: [] for k in FILES}
    manifest = {"arch": {k: v for k, v in a.as_dict().items() if isinstance(v, int)}, "tensors": {}}
    manifest["arch"]["vo_dilations"] = list(a.vo_dilations)
    for i, name in enumerate(names):
        v = ref.tensor(name)
        stage = name.split(".")[0]
        iname = f"/model/{name.replace('.', '/')}"  # graph-side names differ from engine-side names on purpose
        ent = {"file": FILES[stage], "name": iname}
        if name.endswith((".pw1.w", ".pw2.w", ".q.w", ".o.w")) and i % 2 == 0:
            # MatMul-style storage [K][N]: needs "transpose"
            rows = {"pw1": (a_hidden(a, stage), a_dim(a, stage)), "pw2": (a_dim(a, stage), a_hidden(a, stage))}.get(name.split(".")[-2])
            if rows is None:
                rows = (a_dim(a, stage), a_dim(a, stage))
            w = v.reshape(rows)
            per_file[stage].append(ow.tensor(iname, np.ascontiguousarray(w.T)))
            ent["transpose"] = True
        elif i % 7 == 3:
            per_file[stage].append(ow.tensor(iname, v.astype(np.float32), style="packed"))
        elif i % 11 == 5:
            per_file[stage].append(ow.tensor(iname, v.astype(np.float64)))  # exactly representable round trip
        else:
            per_file[stage].append(ow.tensor(iname, v))
        manifest["tensors"][name] = ent
    io = {"dp": (["text_ids", "style_dp", "text_mask"], ["duration"]), "te": (["text_ids", "style_ttl", "text_mask"], ["text_emb"]),
          "ve": (["noisy_latent", "text_emb", "style_ttl", "text_mask", "latent_mask", "total_step", "current_step"], ["denoised_latent"]),
          "vo": (["latent"], ["wav_tts"])}
    for k, fn in FILES.items():
        (tmp / fn).write_bytes(ow.model(per_file[k], [ow.node("Identity", ["x"], ["y"])], *io[k]))
    (tmp / "tts.json").write_text(json.dumps({
        "ae": {"sample_rate": a.sample_rate, "base_chunk_size": a.base_chunk_size},
        "ttl": {"chunk_compress_factor": a.chunk_compress_factor, "latent_dim": a.latent_dim,
                "style_encoder": {"style_token_layer": {"n_style": a.n_style_ttl, "style_value_dim": a.d_style_ttl}},
                "text_encoder": {"proj_out": {"idim": a.te_dim, "odim": a.te_out_dim}}},
        "dp": {"style_encoder": {"style_token_layer": {"n_style": a.n_style_dp, "style_value_dim": a.d_style_dp}}}}))
    (tmp / "unicode_indexer.json").write_text(json.dumps(host.synthetic_indexer().tolist()))
    (tmp / "stn_weight_map.json").write_text(json.dumps(manifest))
    return manifest


def a_dim(a, stage):
    return {"dp": a.dp_dim, "te": a.te_dim, "ve": a.ve_dim, "vo": a.vo_dim}[stage]


def a_hidden(a, stage):
    return {"dp": a.dp_hidden, "te": a.te_hidden, "ve": a.ve_hidden, "vo": a.vo_hidden}[stage]


def test_load_dir_round_trip(tmp_path):
    a = tiny_arch()
    ref = RefModel(a, 7)
    synth = binding.Engine(0, "f32")
    synth.load_synthetic(a, 7)
    build_asset_dir(tmp_path, a, ref, synth)
    s = host.onnx_summary(str(tmp_path / "vocoder.onnx"))
    assert s["inputs"] == ["latent"] and s["outputs"] == ["wav_tts"] and len(s["initializers"]) > 20
    eng = binding.Engine(0, "f32")
    eng.load_dir(str(tmp_path))
    assert eng.param_count == synth.param_count == ref.param_count
    assert eng.arch.ve_dim == a.ve_dim and eng.arch.n_style_ttl == a.n_style_ttl
    ids, mask, sttl, sdp = make_inputs(a, 2, 12, [12, 7], seed=2)
    durs = np.array([0.3, 0.12], np.float32)
    w0, d0 = synth.synthesize(ids, mask, sttl, sdp, 3, 1.05, duration_override=durs, noise_seed=9)
    w1, d1 = eng.synthesize(ids, mask, sttl, sdp, 3, 1.05, duration_override=durs, noise_seed=9)
    assert np.array_equal(w0, w1) and np.array_equal(d0, d1)
    assert np.array_equal(synth.duration(ids, sdp, mask), eng.duration(ids, sdp, mask))


def test_load_dir_error_paths(tmp_path):
    a = tiny_arch()
    ref = RefModel(a, 7)
    eng = binding.Engine(0, "f32")
    with pytest.raises(binding.StnError, match="Failed to open"):
        eng.load_dir(str(tmp_path))  # empty directory: like cpp/helper.cpp:805
    man = build_asset_dir(tmp_path, a, ref, eng)
    (tmp_path / "stn_weight_map.json").rename(tmp_path / "m.json")
    with pytest.raises(binding.StnError, match="no weight manifest.*initializers"):
        eng.load_dir(str(tmp_path))
    broken = dict(man)
    broken["tensors"] = dict(man["tensors"])
    del broken["tensors"]["vo.head.w"]
    (tmp_path / "stn_weight_map.json").write_text(json.dumps(broken))
    with pytest.raises(binding.StnError, match='no entry for tensor "vo.head.w"'):
        eng.load_dir(str(tmp_path))
    broken["tensors"]["vo.head.w"] = {"file": "vocoder.onnx", "name": "/model/vo/head/b"}
    (tmp_path / "stn_weight_map.json").write_text(json.dumps(broken))
    with pytest.raises(binding.StnError, match="elements, descriptor wants"):
        eng.load_dir(str(tmp_path))
