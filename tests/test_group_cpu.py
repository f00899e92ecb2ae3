"""include/stn_group.h without a GPU: the deal is the rule of supertonic_amd/dist.py:shard_by_length (sort by
length, deal round-robin), the block layout that follows from it, and the loud failure of stn_group_create without devices."""
import numpy as np
import pytest

from supertonic_amd import binding
from supertonic_amd.dist import shard_by_length


@pytest.mark.parametrize("B,n", [(1, 1), (7, 2), (128, 8), (1024, 8), (5, 8), (64, 3)])
def test_deal_is_shard_by_length(B, n):
    rng = np.random.default_rng(B * 31 + n)
    lengths = rng.integers(3, 40, B).astype(np.int32)  # many ties: the deal must break them by caller order, like the stable argsort
    rank_of, row_of = binding.group_deal(lengths, n)
    shards = shard_by_length(lengths, n)
    for r in range(n):
        mine = np.where(rank_of == r)[0]
        order = mine[np.argsort(row_of[mine])]
        assert np.array_equal(order, shards[r]), (r, order, shards[r])
        assert np.array_equal(np.sort(row_of[mine]), np.arange(len(mine)))  # rows 0 .. B_r - 1, each once
    # balance: shard sizes differ by at most one, and (lengths sorted, dealt round-robin) so do the shards' longest members by rank order
    sizes = np.bincount(rank_of, minlength=n)
    assert sizes.max() - sizes.min() <= 1 and sizes.sum() == B


def test_deal_rejects_bad_arguments():
    with pytest.raises(binding.StnError):
        binding.group_deal(np.array([3, 4], np.int32), 0)


def test_group_create_without_devices_fails_loudly():
    try:
        n = binding.device_count()
    except binding.StnError:
        n = 0
    with pytest.raises(binding.StnError) as ei:
        binding.Group(n + 1, "bf16")  # one more device than the box has (on a CPU box: one more than none)
    msg = str(ei.value)
    assert ei.value.code == -2 and "visible" in msg and str(n) in msg, msg  # STN_ERR_DEVICE, both numbers named
