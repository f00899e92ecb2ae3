"""In-kernel phase breakdown of the tiled GEMM on the model's shapes (stn_op_gemm_phases): where a launch spends its time.
Ticks are shader-clock cycles of the workgroup's own CU (s_memtime; counters of different XCDs are not comparable, so only
per-workgroup differences are reported).
Usage: python tools/gemm_phases.py [modes...]     (on a GPU box; STN_DEV_SWITCHES=1 STN_GEMM_CFG=<n> forces a tile configuration)
modes: 0 bias+GELU->bf16, 1 residual epilogue, 2 bias only, 3 fp32 store"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from supertonic_amd import binding

eng = binding.Engine(0, "bf16")
modes = [int(a) for a in sys.argv[1:]]
shapes = [("ve.pw1", 9984, 1536, 384, 0), ("ve.pw2", 9984, 384, 1536, 1), ("ve.attn_out", 9984, 384, 384, 1),
          ("vo.pw1", 59904, 2048, 512, 0), ("vo.pw2", 59904, 512, 2048, 1)]
print("cfg", os.environ.get("STN_GEMM_CFG", "auto"))
for name, M, N, K, mode in shapes:
    for md in ([mode] if not modes or mode == 1 else modes):
        ms = eng.op_gemm_bench(M, N, K, md, 50)
        ph = eng.op_gemm_phases(M, N, K, md)
        tot = ph["first_stage"] + ph["k_loop"] + ph["epilogue"]
        print(f"{name:12s} mode {md:3d} M={M} N={N} K={K}: {ms*1e3:7.1f} us/launch ({2.0*M*N*K/ms/1e9:6.0f} TF)  wgs={ph['workgroups']:5d}  "
              f"ticks/wg: first={ph['first_stage']:.0f} kloop={ph['k_loop']:.0f} epi={ph['epilogue']:.0f} total={tot:.0f}", flush=True)
