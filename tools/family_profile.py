"""Per-kernel-family time of one fully profiled C3 step (kernel-exact spans): where a batch's GPU time goes, by stage.
Usage (GPU box): python tools/family_profile.py [--dtype bf16] [--batch 128]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from supertonic_amd import binding, host, workload
from supertonic_amd.arch import default_arch

ap = argparse.ArgumentParser(); ap.add_argument("--dtype", default="bf16"); ap.add_argument("--batch", type=int, default=128)
a_ = ap.parse_args()
arch = default_arch()
texts = workload.utterances(a_.batch, 10, seed=1234)
ids, mask = host.UnicodeProcessor(host.synthetic_indexer())(texts, ["en"] * len(texts))
sttl, sdp = workload.synthetic_styles(arch, list(range(a_.batch)))
durs = workload.forced_durations(texts)
eng = binding.Engine(0, a_.dtype); eng.load_synthetic(arch, 7)
eng.batch_upload(ids, mask, sttl, sdp, duration_override=durs)
for _ in range(3): eng.batch_run(5, 1.05, 1234)
eng.sync()
eng.profile_filter(None); eng.profile_enable(True); eng.profile_reset()
eng.batch_run(5, 1.05, 1234); eng.sync()
st = eng.profile(); eng.profile_enable(False)
tot = sum(v["ms"] for v in st.values())
stage = {}
for k, v in st.items(): stage[k.split(".")[0]] = stage.get(k.split(".")[0], 0.0) + v["ms"]
print("total kernel time %.3f ms; by stage: %s" % (tot, {k: round(v, 3) for k, v in sorted(stage.items(), key=lambda kv: -kv[1])}))
for k, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"{k:24s} {v['ms']:8.3f} ms {100*v['ms']/tot:5.1f} %  launches {v['launches']:4d}  avg {1e3*v['ms']/max(v['launches'],1):7.1f} us")
