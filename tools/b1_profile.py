"""Per-family kernel time of ONE utterance (config C2: batch of one, the fixed 10-word sentence, 5 Euler steps): one fully
event-timed eager step (serialised launches: kernel durations, not latency) next to the replayed latency.
Usage (GPU box): python tools/b1_profile.py [bf16|f32|f16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from supertonic_amd import binding, host, workload
from supertonic_amd.arch import default_arch

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
a = default_arch()
eng = binding.Engine(0, mode)
eng.load_synthetic(a, 7)
up = host.UnicodeProcessor(host.synthetic_indexer())
text = [workload.C1_SENTENCE]
ids, mask = up(text, ["en"])
sttl, sdp = workload.synthetic_styles(a, [0])
durs = workload.forced_durations(text)
eng.batch_upload(ids, mask, sttl, sdp, duration_override=durs, utt_ids=[0])
for _ in range(10):
    eng.batch_run(5, 1.05, 1234)
eng.sync()
lat = []
for _ in range(100):
    t = time.perf_counter(); eng.batch_run(5, 1.05, 1234); eng.sync(); lat.append((time.perf_counter() - t) * 1e3)
print(f"[{mode}] replay p50 {np.percentile(lat, 50):.3f} ms  p90 {np.percentile(lat, 90):.3f} ms  ve rows {eng.ve_rows} vo rows {eng.vo_rows}")
eng.set_graph_mode(False)
lat = []
for _ in range(30):
    t = time.perf_counter(); eng.batch_run(5, 1.05, 1234); eng.sync(); lat.append((time.perf_counter() - t) * 1e3)
print(f"[{mode}] eager  p50 {np.percentile(lat, 50):.3f} ms")
eng.profile_filter(None); eng.profile_sample(1); eng.profile_enable(True); eng.profile_reset()
eng.batch_run(5, 1.05, 1234); eng.sync()
st = eng.profile(); eng.profile_enable(False)
tot = sum(v["ms"] for v in st.values()); n = sum(v["launches"] for v in st.values())
print(f"[{mode}] kernel-time sum {tot:.3f} ms over {n} launches ({tot / n * 1e3:.2f} us each)")
for k, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"  {k:24s} {v['launches']:4d} x {v['ms'] / v['launches'] * 1e3:7.2f} us = {v['ms']:.3f} ms")
