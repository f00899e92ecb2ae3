"""Voice-style loaders on real files in the reference's schema (/root/reference/go/helper.go:87-98,
/root/reference/cpp/helper.cpp:829-897, py/helper.py:340-367): the C++ loader (through stn_load_voice_style) and the Python
loader must both stack the files along dim 0, row-major, in the order given, and fail on a missing file the way the reference
does.  No GPU needed."""
import json
import os

import numpy as np
import pytest

from supertonic_amd import host
from supertonic_amd.tts import load_voice_style


def _write(path, ttl, dp, extra_type=True):
    d = {"style_ttl": {"data": ttl[None].tolist(), "dims": [1, *ttl.shape]},
         "style_dp": {"data": dp[None].tolist(), "dims": [1, *dp.shape]}}
    if extra_type:  # the published files carry a "type" entry too; loaders must ignore it
        d["style_ttl"]["type"] = "float32"
        d["style_dp"]["type"] = "float32"
    with open(path, "w") as f:
        json.dump(d, f)


def _voices(tmp_path, n, d=(50, 256), e=(8, 16)):
    rng = np.random.default_rng(5)
    paths, ttl, dp = [], [], []
    for i in range(n):
        t = rng.standard_normal(d).astype(np.float32)
        q = rng.standard_normal(e).astype(np.float32)
        p = str(tmp_path / f"V{i}.json")
        _write(p, t, q, extra_type=i % 2 == 0)
        paths.append(p); ttl.append(t); dp.append(q)
    return paths, np.stack(ttl), np.stack(dp)


@pytest.mark.parametrize("n", [1, 3])
def test_cpp_and_python_loaders_stack_row_major(tmp_path, n):
    paths, ttl, dp = _voices(tmp_path, n)
    t_c, d_c = host.load_voice_style_native(paths)
    assert t_c.shape == (n, 50, 256) and d_c.shape == (n, 8, 16)
    # float32 -> JSON (repr of the nearest double) -> float32 is exact
    np.testing.assert_array_equal(t_c, ttl)
    np.testing.assert_array_equal(d_c, dp)
    st = load_voice_style(paths)
    np.testing.assert_array_equal(st.ttl, ttl)
    np.testing.assert_array_equal(st.dp, dp)
    # order matters: reversing the paths reverses dim 0
    t_r, d_r = host.load_voice_style_native(paths[::-1])
    np.testing.assert_array_equal(t_r, ttl[::-1])
    np.testing.assert_array_equal(d_r, dp[::-1])


def test_other_dims_come_from_the_files(tmp_path):
    paths, ttl, dp = _voices(tmp_path, 2, d=(6, 32), e=(3, 8))
    t_c, d_c = host.load_voice_style_native(paths)
    assert t_c.shape == (2, 6, 32) and d_c.shape == (2, 3, 8)
    np.testing.assert_array_equal(t_c, ttl)
    np.testing.assert_array_equal(load_voice_style(paths).dp, dp)


def test_missing_file_is_the_reference_error(tmp_path):
    paths, _, _ = _voices(tmp_path, 1)
    missing = str(tmp_path / "nope.json")
    with pytest.raises(OSError) as ei:
        host.load_voice_style_native(paths + [missing])
    assert "Failed to open voice style file: " + missing in str(ei.value)  # cpp/helper.cpp:835,858
    with pytest.raises(FileNotFoundError):
        load_voice_style(paths + [missing])  # py/helper.py:347 opens the file: the same exception class as the reference


def test_data_that_does_not_match_dims_is_rejected(tmp_path):
    p = str(tmp_path / "bad.json")
    with open(p, "w") as f:
        json.dump({"style_ttl": {"data": [[[1.0, 2.0, 3.0]]], "dims": [1, 2, 2]}, "style_dp": {"data": [[[1.0]]], "dims": [1, 1, 1]}}, f)
    with pytest.raises(OSError) as ei:
        host.load_voice_style_native([p])
    assert "does not match dims" in str(ei.value)
    with pytest.raises(ValueError):
        load_voice_style([p])
