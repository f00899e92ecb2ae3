#!/usr/bin/env python3
"""GEMM shape sweep on the GPU (device-resident operands, HIP-event timing) for the shapes of config C3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from supertonic_amd import binding
from supertonic_amd.arch import tiny_arch

SHAPES = [  # (name, M, N, K, mode)
    ("ve.pw1", 9984, 1536, 384, 0), ("ve.pw2", 9984, 384, 1536, 1), ("ve.pw1.pk", 7436, 1536, 384, 0), ("ve.pw2.pk", 7436, 384, 1536, 1), ("ve.q.pk", 7436, 384, 384, 0),
    ("vo.pw1", 59904, 2048, 512, 0), ("vo.pw2", 59904, 512, 2048, 1),
    ("te.pw1", 12032, 1024, 256, 0), ("te.pw2", 12032, 256, 1024, 1),
    ("ve.q", 9984, 384, 384, 0), ("ve.b1.pw1", 78, 1536, 384, 0), ("ve.b1.pw2", 78, 384, 1536, 1),
    ("sq4k", 4096, 4096, 4096, 0),
    ("vo.pw1-noact", 59904, 2048, 512, 2), ("vo.pw1-f32out", 59904, 2048, 512, 3), ("vo.pw1-K2048", 59904, 2048, 2048, 0),
    ("ve.pw1-noact", 9984, 1536, 384, 2), ("ve.pw1-K1536", 9984, 1536, 1536, 0),
]
e = binding.Engine(0, "bf16")
e.load_synthetic(tiny_arch(), 7)
for name, M, N, K, mode in SHAPES:
    for dt in (["bf16", "f32"] if len(sys.argv) > 1 and sys.argv[1] == "all" else ["bf16"]):
        ms = e.op_gemm_bench(M, N, K, mode, 30, dtype=dt)
        print(f"{name:14s} {dt:4s} M={M:6d} N={N:5d} K={K:5d} mode={mode}  {ms*1e3:9.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)
