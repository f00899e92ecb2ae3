"""One C3 batch (128 utterances, bf16, 5 Euler steps, hipGraph replay) split over N engine handles that run concurrently on N
streams, against the same batch on one handle: does concurrency between independent utterance groups return more than the
smaller launches lose?  (Round 2: no — profiles/r02_split_batch_concurrency.txt; re-measured with the K4-split kernels.)
Usage (GPU box): python tools/split_batch.py [split_min_rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from supertonic_amd import binding, host, workload
from supertonic_amd.arch import default_arch

split_min = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
a = default_arch()
texts = workload.utterances(128, 10, seed=1234)
up = host.UnicodeProcessor(host.synthetic_indexer())
order = np.argsort([-len(t) for t in texts], kind="stable")


def run(nsplit, how):
    parts = [order[i::nsplit] for i in range(nsplit)] if how == "interleave" else np.array_split(order, nsplit)
    engs = []
    for idx in parts:
        e = binding.Engine(0, "bf16")
        e.load_synthetic(a, 7)
        e.set_fused_ffn_min_rows(-1, split_min)
        tx = [texts[i] for i in idx]
        ids, mask = up(tx, ["en"] * len(tx))
        sttl, sdp = workload.synthetic_styles(a, idx)
        e.batch_upload(ids, mask, sttl, sdp, duration_override=workload.forced_durations(tx), utt_ids=idx)
        engs.append(e)
    for _ in range(5):
        for e in engs:
            e.batch_run(5, 1.05, 1234)
    for e in engs:
        e.sync()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        for e in engs:
            e.batch_run(5, 1.05, 1234)
        for e in engs:
            e.sync()
    t_sync = (time.perf_counter() - t0) / n * 1e3
    t0 = time.perf_counter()
    for _ in range(n):
        for e in engs:
            e.batch_run(5, 1.05, 1234)
    for e in engs:
        e.sync()
    t_free = (time.perf_counter() - t0) / n * 1e3
    rows = [e.ve_rows for e in engs]
    for e in engs:
        e.close()
    print(f"splits={nsplit} {how:10s}: batch-at-a-time {t_sync:7.3f} ms   free-running {t_free:7.3f} ms   ve rows {rows}", flush=True)


run(1, "interleave")
for ns in (2, 3):
    for how in ("interleave", "bylen"):
        run(ns, how)
run(1, "interleave")
