"""The built-in protobuf reader (supertonic_amd/csrc/host/onnx_reader.cpp) on hand-encoded ONNX files."""
import numpy as np
import pytest

from supertonic_amd import host
import onnx_writer as ow


def test_summary_of_handmade_model(tmp_path):
    rng = np.random.default_rng(0)
    w = rng.standard_normal((4, 3)).astype(np.float32)
    tensors = [ow.tensor("lin.weight", w), ow.tensor("lin.bias", np.arange(4, dtype=np.float32), style="packed"),
               ow.tensor("idx", np.array([3, -1, 70000000000], np.int64), style="packed"),
               ow.tensor("half", np.array([1.5, -2.0], np.float16)), ow.tensor("unp", np.array([0.25, 4.0], np.float32), style="unpacked")]
    nodes = [ow.node("MatMul", ["x", "lin.weight"], ["h"], "mm"), ow.node("Add", ["h", "lin.bias"], ["y"], "add"),
             ow.node("MatMul", ["y", "lin.weight"], ["z"], "mm2")]
    p = tmp_path / "m.onnx"
    p.write_bytes(ow.model(tensors, nodes, inputs=["text_ids", "style_dp", "text_mask"], outputs=["duration"]))
    s = host.onnx_summary(str(p))
    assert s["ir_version"] == 8 and s["producer"] == "stn-tests"
    assert s["inputs"] == ["text_ids", "style_dp", "text_mask"] and s["outputs"] == ["duration"]  # cpp/helper.cpp:512-513 names
    assert s["ops"] == {"MatMul": 2, "Add": 1} and s["n_nodes"] == 3
    names = {t["name"]: t for t in s["initializers"]}
    assert names["lin.weight"]["dims"] == [4, 3] and names["lin.weight"]["dtype"] == 1
    assert names["idx"]["dtype"] == 7 and names["half"]["dtype"] == 10
    assert s["n_params"] == 12 + 4 + 3 + 2 + 2


def test_missing_and_malformed(tmp_path):
    with pytest.raises(OSError, match="Failed to open"):
        host.onnx_summary(str(tmp_path / "nope.onnx"))
    bad = tmp_path / "bad.onnx"
    bad.write_bytes(b"\x3a\xff\xff\xff\xff\x0f" + b"\x00" * 10)  # graph field claiming 4 GiB
    with pytest.raises(OSError, match="overruns|truncated"):
        host.onnx_summary(str(bad))
