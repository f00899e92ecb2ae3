"""Multi-GPU layout of the synthesis path: utterances are independent, so they shard with no data-path
collective; the one exchange is the gather of finished waveforms to rank 0 (RCCL over xGMI when the tensors
live on MI355X; the same code runs on gloo/CPU tensors in the tests)."""
import numpy as np


def shard_by_length(lengths, world_size):
    """Sort utterances by length (descending) and deal them round-robin: balances total frames per rank and
    keeps per-rank padding low.  Returns a list of index arrays, one per rank."""
    order = np.argsort(-np.asarray(lengths), kind="stable")
    return [order[r::world_size] for r in range(world_size)]


def bench_shards(batch, world_size, scaling="weak", mixed=False, words=10, seed=1234):
    """The benchmark's utterances and their deal over the ranks (bench.py).  weak: `batch` utterances per rank (batch * world in
    all); strong: `batch` utterances in all, batch / world per rank (north_star: "a 128-utterance batch at 1, 2, 4 and 8 MI355X").
    Returns (texts_all, shards): shards[r] = indices of rank r's utterances, length-sorted and dealt round-robin."""
    from . import workload
    if scaling not in ("weak", "strong"):
        raise ValueError("scaling must be 'weak' or 'strong'")
    n_total = batch * world_size if scaling == "weak" else batch
    if n_total < world_size:
        raise ValueError(f"strong scaling needs at least one utterance per rank ({n_total} < {world_size})")
    if mixed:
        texts = workload.utterances(n_total, min_words=4, max_words=48, seed=seed)
    else:
        texts = workload.utterances(n_total, words, seed=seed)
    return texts, shard_by_length([len(t) for t in texts], world_size)


class GatherPlan:
    """Shapes of every rank's [B, W] waveform block, exchanged ONCE (tiny all-gather + host read); afterwards a gather is a
    single collective per batch with no host synchronisation.

    The payload is `slots` buffers of [Bmax, Wmax + tail] in the waveform dtype (float32, or int16 PCM: half the bytes over
    xGMI); the tail is 16 bytes per row whose first 4 carry the row's duration as float32 bits, so rows stay 16-byte
    aligned and one collective moves everything.  With two slots the producer fills slot k+1 while slot k is still in
    flight: `launch(k)` starts the gather without blocking and `wait(k)` orders the caller's stream behind it."""

    def __init__(self, wav_shape, device, dtype, dst=0, slots=1):
        import torch
        import torch.distributed as dist
        self.dst, self.rank, self.world = dst, dist.get_rank(), dist.get_world_size()
        hdr = torch.tensor(list(wav_shape), dtype=torch.int64, device=device)
        hdrs = [torch.zeros_like(hdr) for _ in range(self.world)]
        dist.all_gather(hdrs, hdr)
        self.shapes = [(int(h[0]), int(h[1])) for h in hdrs]
        self.Bm, self.Wm = max(s[0] for s in self.shapes), max(s[1] for s in self.shapes)
        isz = torch.empty((), dtype=dtype).element_size()
        self.tail = 16 // isz
        self.Wm = (self.Wm + self.tail - 1) // self.tail * self.tail  # keeps every row (and the tail) 16-byte aligned
        self.stride = self.Wm + self.tail
        self.payload = [torch.zeros((self.Bm, self.stride), dtype=dtype, device=device) for _ in range(slots)]
        self.bufs = [[torch.empty_like(self.payload[0]) for _ in range(self.world)] if self.rank == dst else None
                     for _ in range(slots)]
        self.work = [None] * slots

    @classmethod
    def local(cls, shapes, device, dtype):
        """The ROOT's side of a plan for the given per-rank [B, W] shapes, without a process group: the receive buffers a gather
        into rank 0 would fill, for producers that run one after another in this process (the single-GPU rehearsal of the
        8-shard job, tests/test_gpu_configs.py).  `bufs[0][r]` / `result()` as after a real gather; `wav_ptr(r)` addresses rank
        r's block."""
        import torch
        self = cls.__new__(cls)
        self.dst, self.rank, self.world = 0, 0, len(shapes)
        self.shapes = [(int(b), int(w)) for b, w in shapes]
        self.Bm, self.Wm = max(s[0] for s in self.shapes), max(s[1] for s in self.shapes)
        isz = torch.empty((), dtype=dtype).element_size()
        self.tail = 16 // isz
        self.Wm = (self.Wm + self.tail - 1) // self.tail * self.tail
        self.stride = self.Wm + self.tail
        self.bufs = [[torch.zeros((self.Bm, self.stride), dtype=dtype, device=device) for _ in range(self.world)]]
        self.payload = self.bufs[0]  # wav_ptr(r) / set_durations(d, r) write rank r's block in place
        self.work = [None] * self.world
        return self

    def _dur_view(self, t):
        import torch
        return t[:, self.Wm:].view(torch.float32)[:, 0]

    def wav_ptr(self, k=0):
        """Device address of slot k's first waveform row (rows are `stride` elements apart): a producer writes in place."""
        return self.payload[k].data_ptr()

    def set_durations(self, durations, k=0):
        self._dur_view(self.payload[k])[: durations.shape[0]].copy_(durations, non_blocking=True)

    def launch(self, k=0):
        """Start the gather of slot k behind the work already queued on the current stream; returns at once."""
        import torch.distributed as dist
        import torch
        # moved as raw bytes: neither RCCL/NCCL nor gloo has a 16-bit integer type, and a gather does no arithmetic
        recv = [b.view(torch.uint8) for b in self.bufs[k]] if self.bufs[k] is not None else None
        self.work[k] = dist.gather(self.payload[k].view(torch.uint8), recv, dst=self.dst, async_op=True)

    def wait(self, k=0):
        """Order the current stream (on CPU: the caller) behind slot k's gather; a no-op if none is pending."""
        if self.work[k] is not None:
            self.work[k].wait()
            self.work[k] = None

    def result(self, k=0):
        """On dst, after wait(k): (per-rank wav views, per-rank durations); elsewhere (None, None)."""
        if self.rank != self.dst:
            return None, None
        return ([b[: s[0], : s[1]] for b, s in zip(self.bufs[k], self.shapes)],
                [self._dur_view(b)[: s[0]] for b, s in zip(self.bufs[k], self.shapes)])

    def gather(self, wav, durations, k=0):
        """wav [B, W], durations [B] (float32) on this rank -> on dst: (list of per-rank wav views, list of per-rank durations)."""
        B, W = wav.shape
        self.wait(k)
        self.payload[k][:B, :W].copy_(wav, non_blocking=True)
        self.set_durations(durations, k)
        self.launch(k)
        self.wait(k)
        return self.result(k)


def gather_waveforms(wav, durations, dst=0):
    """One-shot convenience wrapper: plan + gather.  wav [B, W] (float32 or int16 PCM) and durations [B] (torch tensors on this
    rank's device) -> on `dst`: lists of per-rank tensors (row counts and W may differ per rank), elsewhere (None, None).
    A gather into one root is bounded by the root's 7 inbound xGMI links, not by a ring."""
    import torch
    return GatherPlan(tuple(wav.shape), wav.device, wav.dtype, dst).gather(wav, durations.to(torch.float32))
