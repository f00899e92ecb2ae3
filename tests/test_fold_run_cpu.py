"""The run length of the estimator's fold + conv + LayerNorm kernel (stn_dbg_fold_run_frames, host-only): one 1024-thread workgroup fills a CU, so a grid
of B x ceil(longest / run) workgroups is kept within one round where a longer run achieves that — and left alone otherwise."""
import numpy as np
import pytest

from supertonic_amd import binding


def test_the_bench_shape_takes_runs_of_40():
    rng = np.random.default_rng(0)
    lengths = rng.integers(45, 79, size=128)
    lengths[0] = 78
    assert binding.fold_run_frames(lengths, 256) == 40      # 128 x 3 runs of <= 32 = 384 workgroups; 128 x 2 runs of <= 40 = 256 = one round


@pytest.mark.parametrize("B,longest,expect", [
    (128, 64, 0),     # 128 x 2 = 256 with the default already
    (120, 78, 40),    # 360 -> 240
    (128, 90, 48),    # runs of 40 still need 3 per sequence; 48 -> 2
    (128, 100, 0),    # no run length up to 48 fits one round: default (several rounds either way)
    (4, 78, 0),       # a few sequences: the grid is small whatever the run
    (1, 300, 0),
    (200, 40, 40),    # 200 x 2 = 400 -> 200 x 1
    (300, 30, 0),     # 300 workgroups at any run length
])
def test_choice_by_grid(B, longest, expect):
    lengths = np.full(B, longest // 2, np.int32)
    lengths[B // 2] = longest
    assert binding.fold_run_frames(lengths, 256) == expect


def test_other_devices_and_bad_arguments():
    lengths = np.full(64, 70, np.int32)
    assert binding.fold_run_frames(lengths, 256) == 0        # 192 workgroups fit 256 CUs
    assert binding.fold_run_frames(lengths, 128) == 40       # 192 do not fit 128; 128 do
    with pytest.raises(binding.StnError):
        binding.fold_run_frames(lengths, 0)
