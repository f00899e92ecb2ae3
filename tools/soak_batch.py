"""Bit-stability soak of the resident-batch pipeline (stn_batch_*): the same batch synthesised N times, eager and as hipGraph
replays, at the bench shape (C3) and at the mixed-length shape (C4-like), in both 16-bit modes, must return identical waveforms every time.
Run on a GPU box: python tools/soak_batch.py [N]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from supertonic_amd import binding, host, workload
from supertonic_amd.arch import default_arch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
arch = default_arch()
up = host.UnicodeProcessor(host.synthetic_indexer())
bad = 0
for dtype in ("bf16", "f16"):
    for name, kw in (("C3", dict(min_words=10, max_words=10, seed=1234)), ("mixed", dict(min_words=4, max_words=48, seed=101))):
        texts = workload.utterances(128, **kw)
        ids, mask = up(texts, ["en"] * 128)
        sttl, sdp = workload.synthetic_styles(arch, list(range(128)))
        durs = workload.forced_durations(texts)
        for graph in (0, 1):
            eng = binding.Engine(0, dtype); eng.load_synthetic(arch, 7); eng.set_graph_mode(graph)
            eng.batch_upload(ids, mask, sttl, sdp, duration_override=durs)
            ref = None
            for it in range(N):
                eng.batch_run(5, 1.05, 1234)
                wav, _ = eng.batch_fetch()
                if ref is None: ref = wav.copy()
                elif not np.array_equal(ref, wav):
                    bad += 1
                    d = np.argwhere(ref != wav)
                    print(f"  {dtype} {name} graph={graph} run {it}: {len(d)} samples differ, utterances {np.unique(d[:, 0])[:8]}")
            print(f"{dtype} {name} graph={graph}: {N} runs, wav {ref.shape}, finite {bool(np.isfinite(ref).all())}", flush=True)
            eng.close()
print("SOAK", "FAILED" if bad else "OK", bad)
sys.exit(1 if bad else 0)
