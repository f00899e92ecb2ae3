"""K4 / K4-split against the two launches they replace, on the model's shapes (stn_op_ffn_bench, stn_op_block_bench):
device-resident random operands, HIP-event timing over `iters` calls, plus the fused kernels' in-kernel phase stamps
(shader-clock cycles per workgroup).  Usage (on a GPU box): python tools/ffn_bench.py [iters] [sweep|split]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from supertonic_amd import binding

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
what = sys.argv[2] if len(sys.argv) > 2 else ""
eng = binding.Engine(0, "bf16")
if what == "splitvar":  # one line for the timing-variants build (STN_LIB=build/var/libstn.so STN_DEV_SWITCHES=1 STN_FFN_VAR=<n>)
    f = [eng.op_ffn_bench(7436, 384, 1536, 2, iters) for _ in range(3)]
    best = min(f, key=lambda r: r["ms"])
    print(f"var={os.environ.get('STN_FFN_VAR', '0')} ve M=7436 K4-split {best['ms']*1e3:7.1f} us cycles/wg: first={best['first_stage']:.0f} loop={best['tile_loop']:.0f} epi={best['epilogue']:.0f}", flush=True)
    sys.exit(0)
if what == "splitm":  # few rows: which split pays (STN_DEV_SWITCHES=1 STN_FFN_SPLIT_S=<4|8|12|24> forces one; unset: ffn_split_choose)
    tag = os.environ.get("STN_FFN_SPLIT_S", "auto")
    for B in (1, 2, 4, 8, 16, 32, 48, 64, 96, 128):
        a = min((eng.op_block_bench(B, 58, 384, 1536, 5, 2, 0, iters) for _ in range(2)), key=lambda r: r["ms"])
        b = min((eng.op_block_bench(B, 58, 384, 1536, 5, 2, 2, iters) for _ in range(2)), key=lambda r: r["ms"])
        k = min((eng.op_ffn_bench(B * 58, 384, 1536, 2, iters) for _ in range(2)), key=lambda r: r["ms"])
        print(f"S={tag:4s} B={B:4d} rows={B*58:5d}: three launches {a['ms']*1e3:6.1f} us (dwconv_ln {a['conv_ms']*1e3:5.1f})   fold_dwconv_ln + K4-split {b['ms']*1e3:6.1f} us "
              f"(fold {b['conv_ms']*1e3:5.1f}, K4-split {k['ms']*1e3:5.1f}: wgs={k['workgroups']} first={k['first_stage']:.0f} loop={k['tile_loop']:.0f} epi={k['epilogue']:.0f})", flush=True)
    sys.exit(0)
if what == "split":
    # the estimator's block: pointwise pair alone (two launches / K4-split), and the whole block chain with its conv kernel
    for M in (58, 464, 1024, 2048, 4096, 7436, 9984, 14872, 20000, 29744):
        f, u = [], []
        for rnd in range(3):
            f.append(eng.op_ffn_bench(M, 384, 1536, 2, iters))
            u.append(eng.op_ffn_bench(M, 384, 1536, 0, iters)["ms"])
        best = min(f, key=lambda r: r["ms"])
        fl = 4.0 * M * 384 * 1536
        print(f"ve M={M:6d}: K4-split {best['ms']*1e3:7.1f} us ({fl/best['ms']/1e9:6.0f} TF)  two launches {min(u)*1e3:7.1f} us ({fl/min(u)/1e9:6.0f} TF)  "
              f"wgs={best['workgroups']} cycles/wg: first={best['first_stage']:.0f} loop={best['tile_loop']:.0f} epi={best['epilogue']:.0f}", flush=True)
    for B, L in ((128, 58), (16, 58), (1, 58), (128, 150)):
        for dil in (1, 2, 4, 8):
            a = [eng.op_block_bench(B, L, 384, 1536, 5, dil, 0, iters) for _ in range(2)]
            b = [eng.op_block_bench(B, L, 384, 1536, 5, dil, 2, iters) for _ in range(2)]
            a = min(a, key=lambda r: r["ms"]); b = min(b, key=lambda r: r["ms"])
            print(f"block B={B:4d} L={L:4d} rows={B*L:6d} dil={dil}: three launches {a['ms']*1e3:6.1f} us (dwconv_ln {a['conv_ms']*1e3:5.1f})   "
                  f"fold_dwconv_ln + K4-split {b['ms']*1e3:6.1f} us (fold_dwconv_ln {b['conv_ms']*1e3:5.1f}; cycles/wg p1={b['fold_phase1']:.0f} bar={b['fold_barrier']:.0f} "
                  f"p2={b['fold_phase2']:.0f} span={b['fold_span']:.0f})", flush=True)
    sys.exit(0)
shapes = [("vo  (C3 dense)", 59904, 512, 2048), ("vo  (half)", 29952, 512, 2048), ("ve  (C3 packed)", 7436, 384, 1536),
          ("ve  (padded)", 9984, 384, 1536), ("ve  (8 batches)", 59488, 384, 1536)]
if what == "sweep":  # where the fused kernel starts to pay: rows sweep at the vocoder's width
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
