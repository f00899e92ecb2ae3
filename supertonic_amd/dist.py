"""Multi-GPU layout of the synthesis path: utterances are independent, so they shard with no data-path
collective; the one exchange is the gather of finished waveforms to rank 0 (RCCL over xGMI when the tensors
live on MI355X; the same code runs on gloo/CPU tensors in the tests)."""
import numpy as np


def shard_by_length(lengths, world_size):
    """Sort utterances by length (descending) and deal them round-robin: balances total frames per rank and
    keeps per-rank padding low.  Returns a list of index arrays, one per rank."""
    order = np.argsort(-np.asarray(lengths), kind="stable")
    return [order[r::world_size] for r in range(world_size)]


def gather_waveforms(wav, durations, dst=0):
    """wav [B, W] and durations [B] (torch tensors on this rank's device) -> on `dst`: lists of per-rank tensors
    (row counts and W may differ per rank), elsewhere None.  One all_gather of the tiny shape header, then
    point-to-point-shaped gather of equal-size padded payloads (ring collectives are per-link bound on xGMI; a
    gather into one root is 7 independent inbound links)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    hdr = torch.tensor([wav.shape[0], wav.shape[1]], dtype=torch.int64, device=wav.device)
    hdrs = [torch.zeros_like(hdr) for _ in range(world)]
    dist.all_gather(hdrs, hdr)
    shapes = [(int(h[0]), int(h[1])) for h in hdrs]
    Bm, Wm = max(s[0] for s in shapes), max(s[1] for s in shapes)
    payload = torch.zeros((Bm, Wm + 1), dtype=wav.dtype, device=wav.device)  # last column carries the duration
    payload[: wav.shape[0], : wav.shape[1]] = wav
    payload[: wav.shape[0], Wm] = durations.to(wav.dtype)
    bufs = [torch.empty_like(payload) for _ in range(world)] if rank == dst else None
    dist.gather(payload, bufs, dst=dst)
    if rank != dst:
        return None, None
    wavs = [b[: s[0], : s[1]] for b, s in zip(bufs, shapes)]
    durs = [b[: s[0], Wm] for b, s in zip(bufs, shapes)]
    return wavs, durs
