"""Multi-GPU layout of the synthesis path: utterances are independent, so they shard with no data-path
collective; the one exchange is the gather of finished waveforms to rank 0 (RCCL over xGMI when the tensors
live on MI355X; the same code runs on gloo/CPU tensors in the tests)."""
import numpy as np


def shard_by_length(lengths, world_size):
    """Sort utterances by length (descending) and deal them round-robin: balances total frames per rank and
    keeps per-rank padding low.  Returns a list of index arrays, one per rank."""
    order = np.argsort(-np.asarray(lengths), kind="stable")
    return [order[r::world_size] for r in range(world_size)]


class GatherPlan:
    """Shapes of every rank's [B, W] waveform block, exchanged ONCE (tiny all-gather + host read); afterwards
    `gather` is a single collective per batch with no host synchronisation."""

    def __init__(self, wav_shape, device, dtype, dst=0):
        import torch
        import torch.distributed as dist
        self.dst, self.rank, self.world = dst, dist.get_rank(), dist.get_world_size()
        hdr = torch.tensor(list(wav_shape), dtype=torch.int64, device=device)
        hdrs = [torch.zeros_like(hdr) for _ in range(self.world)]
        dist.all_gather(hdrs, hdr)
        self.shapes = [(int(h[0]), int(h[1])) for h in hdrs]
        self.Bm, self.Wm = max(s[0] for s in self.shapes), max(s[1] for s in self.shapes)
        self.payload = torch.zeros((self.Bm, self.Wm + 1), dtype=dtype, device=device)  # last column carries the duration
        self.bufs = [torch.empty_like(self.payload) for _ in range(self.world)] if self.rank == dst else None

    def gather(self, wav, durations):
        """wav [B, W], durations [B] on this rank -> on dst: (list of per-rank wav views, list of per-rank durations)."""
        import torch.distributed as dist
        B, W = wav.shape
        self.payload[:B, :W].copy_(wav, non_blocking=True)
        self.payload[:B, self.Wm].copy_(durations, non_blocking=True)
        dist.gather(self.payload, self.bufs, dst=self.dst)
        if self.rank != self.dst:
            return None, None
        return ([b[: s[0], : s[1]] for b, s in zip(self.bufs, self.shapes)],
                [b[: s[0], self.Wm] for b, s in zip(self.bufs, self.shapes)])


def gather_waveforms(wav, durations, dst=0):
    """One-shot convenience wrapper: plan + gather.  wav [B, W] and durations [B] (torch tensors on this rank's device) ->
    on `dst`: lists of per-rank tensors (row counts and W may differ per rank), elsewhere (None, None).
    A gather into one root is bounded by the root's 7 inbound xGMI links, not by a ring."""
    return GatherPlan(tuple(wav.shape), wav.device, wav.dtype, dst).gather(wav, durations.to(wav.dtype))
