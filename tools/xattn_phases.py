"""Phase stamps of ONE workgroup of the one-launch cross-attention kernel (kernels_xattn.hip) inside a C3 batch (measurement build:
hipcc -DXA_STAMPS on kernels_xattn.hip, STN_LIB=<that library>).  The stamped launch is the last one of the run (a style block)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from supertonic_amd import binding, host, workload
from supertonic_amd.arch import default_arch

lib = binding.load()
if not hasattr(lib, "stn_dbg_xa"):
    sys.exit("build kernels_xattn.hip with -DXA_STAMPS and point STN_LIB at the library")
a = default_arch()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
texts = workload.utterances(n, 10, seed=1234)
up = host.UnicodeProcessor(host.synthetic_indexer())
ids, mask = up(texts, ["en"] * n)
sttl, sdp = workload.synthetic_styles(a, np.arange(n))
eng = binding.Engine(0, "bf16")
eng.load_synthetic(a, 7)
eng.set_fused_xattn(True)
eng.set_graph_mode(False)
eng.batch_upload(ids, mask, sttl, sdp, duration_override=workload.forced_durations(texts), utt_ids=np.arange(n))
for _ in range(3):
    eng.batch_run(5, 1.05, 1234)
eng.sync()
ts = (ctypes.c_ulonglong * 16)()
lib.stn_dbg_xa(ts)
t = list(ts)
names = [(0, 1, "fold + LayerNorm"), (1, 2, "q projection (MFMA loop)"), (2, 3, "q write-back"), (3, 4, "rotation + scale"), (4, 5, "barrier"),
         (5, 6, "round 0: K/V staging"), (6, 7, "round 0: attention"), (7, 8, "round 1: K/V staging"), (8, 13, "round 1: attention"),
         (13, 14, "output projection (MFMA loop)"), (14, 15, "residual update")]
print({nm: t[b] - t[a0] for a0, b, nm in names}, "total", t[15] - t[0], flush=True)
