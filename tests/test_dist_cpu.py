"""The N > 1 path on CPU: world_size-2 gloo run of the sharding + waveform gather used by bench.py
(on MI355X the same code runs over RCCL), plus the sharding invariants."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from supertonic_amd import workload
from supertonic_amd.dist import GatherPlan, bench_shards, gather_waveforms, shard_by_length


def test_shard_by_length_balances_and_partitions():
    texts = workload.utterances(1024, min_words=4, max_words=48, seed=1234)  # C4: mixed lengths
    lens = np.array([len(t) for t in texts])
    shards = shard_by_length(lens, 8)
    allidx = np.sort(np.concatenate(shards))
    assert np.array_equal(allidx, np.arange(1024))  # a partition: nothing lost, nothing duplicated
    assert all(len(s) == 128 for s in shards)
    tot = np.array([lens[s].sum() for s in shards])
    assert tot.max() / tot.min() < 1.02  # balanced total work
    # length-sorted dealing keeps per-rank padding close to the global sorted order
    assert all(np.all(np.diff(lens[s]) <= 0) for s in shards)


def test_bench_shards_weak_and_strong():
    """bench.py --scaling weak|strong: 128 utterances per rank, or 128 in all (north_star's 128-utterance batch at 1/2/4/8 GPUs)."""
    for world in (1, 2, 4, 8):
        tw, sw = bench_shards(128, world, "weak")
        ts, ss = bench_shards(128, world, "strong")
        assert len(tw) == 128 * world and all(len(x) == 128 for x in sw)
        assert len(ts) == 128 and all(len(x) == 128 // world for x in ss)
        for texts, sh in ((tw, sw), (ts, ss)):
            assert np.array_equal(np.sort(np.concatenate(sh)), np.arange(len(texts)))
            tot = np.array([sum(len(texts[i]) for i in x) for x in sh])
            assert tot.max() / tot.min() < 1.06
    assert bench_shards(128, 1, "weak")[0] == bench_shards(128, 1, "strong")[0]  # one GPU: the same job either way
    import pytest
    with pytest.raises(ValueError):
        bench_shards(4, 8, "strong")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _worker_body(rank, world, q)
    except Exception as e:  # report instead of leaving the parent to time out on the queue
        q.put("rank %d: %r" % (rank, e))
    finally:
        dist.destroy_process_group()


def _worker_body(rank, world, q):
    if True:
        # strong scaling as bench.py does it: every rank derives the same deal and takes its own share
        texts, shards = bench_shards(128, world, "strong")
        got = [None] * world
        dist.all_gather_object(got, [int(i) for i in shards[rank]])
        if sorted(sum(got, [])) != list(range(128)) or any(len(g) != 128 // world for g in got):
            q.put("strong-scaling shards do not partition the batch")
            return
        B, W = (3, 40) if rank == 0 else (2, 56)  # ragged: row counts and lengths differ per rank
        wav = torch.arange(B * W, dtype=torch.float32).reshape(B, W) + 1000 * rank
        dur = torch.arange(B, dtype=torch.float32) + 10 * rank
        wavs, durs = gather_waveforms(wav, dur, dst=0)
        plan = GatherPlan((B, W), wav.device, wav.dtype, dst=0)  # the steady-state form used by bench.py: plan once, gather often
        for rep in range(2):
            w2, d2 = plan.gather(wav + rep, dur)
            if rank == 0 and not all(torch.equal(a, b + rep) for a, b in zip(w2, wavs)):
                q.put(False)
                return
        # int16 PCM payload, two slots in flight (what bench.py does on RCCL: fill slot k+1 while slot k travels)
        pcm = (torch.arange(B * W, dtype=torch.int32).reshape(B, W) % 30000 - 15000 + rank).to(torch.int16)
        plan2 = GatherPlan((B, W), pcm.device, torch.int16, dst=0, slots=2)
        for rep in range(4):
            k = rep & 1
            plan2.wait(k)
            plan2.payload[k][:B, :W].copy_(pcm + rep)
            plan2.set_durations(dur + rep, k)
            plan2.launch(k)
        for k in (0, 1):
            plan2.wait(k)
            w3, d3 = plan2.result(k)
            if rank == 0:
                for r in range(world):
                    b, w = (3, 40) if r == 0 else (2, 56)
                    exp = (torch.arange(b * w, dtype=torch.int32).reshape(b, w) % 30000 - 15000 + r).to(torch.int16) + (2 + k)
                    if not (torch.equal(w3[r], exp) and torch.equal(d3[r], torch.arange(b, dtype=torch.float32) + 10 * r + 2 + k)):
                        q.put(False)
                        return
        if rank == 0:
            ok = len(wavs) == world
            for r in range(world):
                b, w = (3, 40) if r == 0 else (2, 56)
                exp = torch.arange(b * w, dtype=torch.float32).reshape(b, w) + 1000 * r
                ok = ok and torch.equal(wavs[r], exp) and torch.equal(durs[r], torch.arange(b, dtype=torch.float32) + 10 * r)
            q.put(bool(ok))
        else:
            q.put(wavs is None and durs is None)


def test_gather_waveforms_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r is True for r in res), res
    assert all(p.exitcode == 0 for p in procs)


def test_workload_is_deterministic():
    a, b = workload.utterances(16, 10), workload.utterances(16, 10)
    assert a == b and all(t[0].isupper() and t.endswith(".") and len(t.split()) == 10 for t in a)
    d = workload.forced_durations([workload.C1_SENTENCE])
    assert abs(float(d[0]) - 53 / 15.0) < 1e-6  # 53 chars -> 3.53 s before /speed


def test_gather_plan_local_is_the_roots_receive_side():
    """GatherPlan.local: the root's buffers for given per-rank shapes with no process group (what the single-GPU rehearsal of the
    8-shard job fills rank by rank): same row stride / duration tail / result() views as after a real gather."""
    import torch
    from supertonic_amd.dist import GatherPlan
    shapes = [(3, 40), (2, 56), (1, 8)]
    plan = GatherPlan.local(shapes, torch.device("cpu"), torch.int16)
    assert plan.world == 3 and plan.Bm == 3 and plan.Wm == 56 and plan.stride == 56 + 8 and plan.stride * 2 % 16 == 0
    for r, (b, w) in enumerate(shapes):
        blk = plan.payload[r]
        blk[:b, :w] = (torch.arange(b * w, dtype=torch.int32).reshape(b, w) % 1000 + 7 * r).to(torch.int16)
        plan.set_durations(torch.arange(b, dtype=torch.float32) + 10 * r, r)
    wavs, durs = plan.result(0)
    for r, (b, w) in enumerate(shapes):
        exp = (torch.arange(b * w, dtype=torch.int32).reshape(b, w) % 1000 + 7 * r).to(torch.int16)
        assert wavs[r].shape == (b, w) and torch.equal(wavs[r], exp)
        assert torch.equal(durs[r], torch.arange(b, dtype=torch.float32) + 10 * r)
