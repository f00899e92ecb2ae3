"""stn_load_dir without a manifest, host side (no device): the loader has to find the engine's layout in the graphs' nodes —
descriptor from the weight shapes / Conv attributes / Reshape constants, tensors bound by position and role — and say precisely
where a graph stops being that layout.  Stands in for what Ort::Session does with the files (/root/reference/cpp/helper.cpp:784-795);
the graphs are emitted by tests/onnx_graphs.py from the oracle's weights (the published graphs are not available offline, so this
pins the loader against the layout of include/stn_arch.h, not against the published files: parity unpinned for those)."""
import numpy as np
import pytest

from oracle.neural_ref import RefModel
from supertonic_amd import host
from supertonic_amd.arch import tiny_arch
from onnx_graphs import build_graph_dir

ARCH_FIELDS = ["vocab_size", "te_dim", "te_hidden", "te_kernel", "te_conv_blocks", "te_attn_blocks", "te_heads", "te_ffn", "te_style_blocks",
               "te_out_dim", "dp_dim", "dp_hidden", "dp_kernel", "dp_conv_blocks", "dp_heads", "ve_dim", "ve_hidden", "ve_kernel",
               "ve_main_blocks", "ve_dilated", "ve_tail_blocks", "ve_heads", "ve_time_dim", "vo_dim", "vo_hidden", "vo_kernel", "vo_blocks",
               "vo_in_kernel", "d_style_ttl", "d_style_dp", "n_style_ttl", "n_style_dp", "sample_rate", "base_chunk_size", "latent_dim"]


def _zeros(a):
    ref = RefModel(a, 7)
    return ref.tensor


def test_descriptor_and_binding_come_out_of_the_graphs(tmp_path):
    a = tiny_arch()
    ref = RefModel(a, 7)
    build_graph_dir(tmp_path, a, ref.tensor)
    got = host.bind_graphs(str(tmp_path))
    for f in ARCH_FIELDS:
        assert got["arch"][f] == getattr(a, f), f
    assert got["arch"]["vo_dilations"] == list(a.vo_dilations)[:a.vo_blocks]
    assert set(got["tensors"]) == set(ref.tensor_names())
    assert got["notes"] == "" and not any(e["zeros"] for e in got["tensors"].values())  # every bias found (dp.fc2's has ONE element)
    # the four projection encodings all occur, and exactly the [K][N]-stored ones are marked for transposition
    tr = [n for n, e in got["tensors"].items() if e["transpose"]]
    assert tr and all(n.endswith(".w") for n in tr)
    froms = " ".join(e["from"] for e in got["tensors"].values())
    for op in ("Conv", "MatMul", "Gemm", "Gather", "LayerNormalization", "Mul"):
        assert f" {op} '" in froms, op


def test_another_depth_and_width_is_just_another_descriptor(tmp_path):
    a = tiny_arch()
    a.vo_blocks, a.ve_main_blocks, a.ve_dilated, a.ve_tail_blocks, a.te_attn_blocks, a.te_style_blocks = 2, 1, 3, 2, 1, 2
    a.vo_dim, a.vo_hidden, a.dp_conv_blocks, a.vo_kernel = 48, 96, 1, 5
    for i, d in enumerate((1, 3)):
        a.vo_dilations[i] = d
    ref = RefModel(a, 3)
    build_graph_dir(tmp_path, a, ref.tensor)
    got = host.bind_graphs(str(tmp_path))
    for f in ARCH_FIELDS:
        assert got["arch"][f] == getattr(a, f), f
    assert got["arch"]["vo_dilations"] == [1, 3]
    assert set(got["tensors"]) == set(ref.tensor_names()) and not any(e["zeros"] for e in got["tensors"].values())


def test_estimator_without_dilated_blocks_or_tail(tmp_path):
    a = tiny_arch()
    a.ve_dilated, a.ve_tail_blocks, a.ve_main_blocks = 0, 0, 3
    build_graph_dir(tmp_path, a, _zeros(a))
    got = host.bind_graphs(str(tmp_path))["arch"]
    assert (got["ve_dilated"], got["ve_tail_blocks"], got["ve_main_blocks"]) == (0, 0, 3)


def test_heads_missing_from_the_graph_are_reported_not_guessed(tmp_path):
    a = tiny_arch()
    build_graph_dir(tmp_path, a, _zeros(a), with_heads=False)
    got = host.bind_graphs(str(tmp_path))
    for f in ("te_heads", "dp_heads", "ve_heads"):
        assert f in got["notes"]


@pytest.mark.parametrize("breaks,needle", [
    ({"vo.blk1": "width"}, r"vocoder\.onnx.*vo\.blk1\.dw = depthwise Conv over 64 channels.*the graph has depthwise Conv 72 <- 72.*node #\d+ Conv.*\[72,1,7\].*vo_blocks=1"),
    ({"te.conv1": "no_gamma"}, r"text_encoder\.onnx.*te\.conv1\.gamma = per-channel scale \(Mul\) over 64 channels; the graph has LayerNormalization"),
    ({"ve.m0.dil1": "no_gamma"}, r"vector_estimator\.onnx.*ve\.m0\.dil1\.gamma = per-channel scale.*ve_hidden=192.*ve_dilated=0"),
    ({"ve.m1.cn_a": "batchnorm"}, r"vector_estimator\.onnx.*ve\.m1\.cn_a\.pw1.*the graph has unrecognised weighted operator.*BatchNormalization"),
])
def test_a_graph_that_is_not_the_layout_fails_with_the_first_node_that_does_not_fit(tmp_path, breaks, needle):
    a = tiny_arch()
    build_graph_dir(tmp_path, a, _zeros(a), breaks=breaks)
    with pytest.raises(OSError, match=needle):
        host.bind_graphs(str(tmp_path))


def test_tts_json_must_agree_with_the_graphs(tmp_path):
    a = tiny_arch()
    build_graph_dir(tmp_path, a, _zeros(a), tts_overrides={("ae", "base_chunk_size"): 256})
    with pytest.raises(OSError, match=r"base_chunk_size \(rows of vo\.head\) = 512 by the graph's weight shapes, but tts\.json says 256"):
        host.bind_graphs(str(tmp_path))
    build_graph_dir(tmp_path, a, _zeros(a), tts_overrides={("ttl", "latent_dim"): 16})
    with pytest.raises(OSError, match=r"latent_dim \* chunk_compress_factor"):
        host.bind_graphs(str(tmp_path))


def test_graph_io_names_are_checked(tmp_path):
    a = tiny_arch()
    build_graph_dir(tmp_path, a, _zeros(a), io_overrides={"vo": (["z"], ["wav_tts"])})
    with pytest.raises(OSError, match=r"vocoder\.onnx: graph inputs are \{z\}, the host feeds \{latent\}"):
        host.bind_graphs(str(tmp_path))
