"""Phase stamps of fold_dwconv_ln (K4-split's successor) at a few batch sizes and dilations: cycles of phase 1 (loads + fold into the LDS image),
the barrier, phase 2 (conv + LayerNorm + stores), next to the block's chain time.  Usage (GPU box): python tools/fold_phases.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from supertonic_amd import binding

eng = binding.Engine(0, "bf16")
for B in (1, 2, 8, 128):
    for dil in (1, 8):
        r = min((eng.op_block_bench(B, 58, 384, 1536, 5, dil, 2, 30) for _ in range(3)), key=lambda r: r["ms"])
        print(f"B={B:4d} dil={dil}: block {r['ms']*1e3:6.1f} us, fold_dwconv_ln {r['conv_ms']*1e3:5.1f} us; cycles phase 1 {r['fold_phase1']:.0f}, barrier {r['fold_barrier']:.0f}, "
              f"phase 2 {r['fold_phase2']:.0f}", flush=True)
