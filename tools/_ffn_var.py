import os, sys
sys.path.insert(0, "/root/repo")
from supertonic_amd import binding
eng = binding.Engine(0, "bf16")
M, C, I = int(os.environ.get("FM", 59904)), int(os.environ.get("FC", 512)), int(os.environ.get("FI", 2048))
for rnd in range(2):
    r = eng.op_ffn_bench(M, C, I, True, 20)
    print(os.environ.get("STN_FFN_VAR", "0"), M, C, I, f"fused {r['ms']*1e3:7.1f} us  first={r['first_stage']:.0f} loop={r['tile_loop']:.0f} epi={r['epilogue']:.0f}", flush=True)
