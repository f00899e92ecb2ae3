"""Phase stamps of the head-split cross-attention kernel (kernels_xattn_hs.hip) inside a C3 batch: the stamped launch is the last one of
the run (a style block of the last Euler step).  Usage (GPU box): python tools/xattn_hs_phases.py [n_utterances]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from supertonic_amd import binding, host, workload
from supertonic_amd.arch import default_arch

a = default_arch()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
texts = workload.utterances(n, 10, seed=1234)
ids, mask = host.UnicodeProcessor(host.synthetic_indexer())(texts, ["en"] * n)
sttl, sdp = workload.synthetic_styles(a, np.arange(n))
eng = binding.Engine(0, "bf16")
eng.load_synthetic(a, 7)
eng.set_fused_xattn(1)
eng.set_graph_mode(False)
eng.xattn_hs_stamps_enable(True)
eng.batch_upload(ids, mask, sttl, sdp, duration_override=workload.forced_durations(texts), utt_ids=np.arange(n))
for _ in range(3):
    eng.batch_run(5, 1.05, 1234)
eng.sync()
ts = eng.xattn_hs_stamps().astype(np.int64)
live = ts[ts[:, 0] > 0]
names = ["Wq in LDS (loads issued: Wq, xn, K/V)", "q projection + rotation", "K/V commit, Wo loads issued", "attention", "Wo commit", "output projection + stores"]
d = np.diff(live[:, :7], axis=1)
print(f"{len(live)} workgroups, tiles per workgroup {np.bincount(live[:, 7])}")
for i, nm in enumerate(names):
    print(f"  {nm:42s} mean {d[:, i].mean():8.0f}  min {d[:, i].min():7d}  max {d[:, i].max():7d} cycles")
tot = live[:, 6] - live[:, 0]
print(f"  total per workgroup: mean {tot.mean():.0f}  max {tot.max()}; launch span (first entry -> last end) {live[:, 6].max() - live[:, 0].min()} cycles")
