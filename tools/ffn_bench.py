"""K4 against the two launches it replaces, on the model's shapes (stn_op_ffn_bench): device-resident random operands,
HIP-event timing over `iters` calls, plus the fused kernel's in-kernel phase stamps (shader-clock cycles per workgroup).
Usage (on a GPU box): python tools/ffn_bench.py [iters] [sweep]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from supertonic_amd import binding

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
eng = binding.Engine(0, "bf16")
shapes = [("vo  (C3 dense)", 59904, 512, 2048), ("vo  (half)", 29952, 512, 2048), ("ve  (C3 packed)", 7436, 384, 1536),
          ("ve  (padded)", 9984, 384, 1536), ("ve  (8 batches)", 59488, 384, 1536)]
if len(sys.argv) > 2:  # where the fused kernel starts to pay: rows sweep at the vocoder's width
    shapes = [(f"vo M={m}", m, 512, 2048) for m in (294, 2048, 4096, 8192, 12288, 16384, 20480, 24576, 28672, 32768, 36864, 49152)]
for name, M, C, I in shapes:
    # interleave the two arms (rule: A/B in one process, alternating)
    f, u = [], []
    for rnd in range(3):
        rf = eng.op_ffn_bench(M, C, I, True, iters)
        ru = eng.op_ffn_bench(M, C, I, False, iters)
        f.append(rf); u.append(ru["ms"])
    best = min(f, key=lambda r: r["ms"])
    fl = 4.0 * M * C * I
    print(f"{name:16s} M={M:6d} C={C} I={I}: fused {min(r['ms'] for r in f)*1e3:7.1f} us ({fl/min(r['ms'] for r in f)/1e9:6.0f} TF)  "
          f"two launches {min(u)*1e3:7.1f} us ({fl/min(u)/1e9:6.0f} TF)  wgs={best['workgroups']} "
          f"cycles/wg: first={best['first_stage']:.0f} loop={best['tile_loop']:.0f} epi={best['epilogue']:.0f}", flush=True)
