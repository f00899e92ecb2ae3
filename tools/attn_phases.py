"""Phase stamps of one workgroup of the MFMA attention kernel on the estimator's cross-attention shapes (measurement build:
`make EXTRA=-DSTN_ATTN_STAMPS`).  Shader-clock cycles between: entry, Q staged, K staged, V staged + barrier, max pass, PV pass, stores.
Since round 3 the loads of Q, K and V are all issued before the first LDS store (one key chunk): "Q staged" holds the wait for ALL of them plus
the Q rotation and stores, "K staged" / "V staged" only their arithmetic and LDS stores."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from supertonic_amd import binding

eng = binding.Engine(0, "bf16")
lib = binding.load()
if not hasattr(lib, "stn_dbg_attn_ts"):
    sys.exit("build with: make EXTRA=-DSTN_ATTN_STAMPS")
rng = np.random.default_rng(0)
for name, Lq, Lk, rope in (("text (LARoPE)", 78, 94, 1), ("style", 78, 50, -1)):
    B, H, dh = 128, 4, 96
    q = rng.standard_normal((B, Lq, H * dh)).astype(np.float32)
    k = rng.standard_normal((B, Lk, H * dh)).astype(np.float32)
    v = rng.standard_normal((B, Lk, H * dh)).astype(np.float32)
    qlen = np.full(B, 58, np.int32); klen = np.full(B, min(Lk, 70), np.int32)
    for _ in range(3):
        eng.op_attention(q, k, v, H, qlen, klen, rope)
    ts = (ctypes.c_ulonglong * 8)()
    lib.stn_dbg_attn_ts(ts)
    t = list(ts)
    names = ["Q staged", "K staged", "V staged+barrier", "max pass", "PV pass", "stores"]
    print(name, {n: t[i + 1] - t[i] for i, n in enumerate(names)}, "total", t[6] - t[0], flush=True)
