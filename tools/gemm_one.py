#!/usr/bin/env python3
"""One GEMM shape, a few launches: for rocprofv3 --pmc runs.  usage: gemm_one.py M N K mode iters"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from supertonic_amd import binding
from supertonic_amd.arch import tiny_arch
M, N, K, mode, iters = [int(a) for a in sys.argv[1:6]]
e = binding.Engine(0, "bf16")
e.load_synthetic(tiny_arch(), 7)
ms = e.op_gemm_bench(M, N, K, mode, iters)
print(f"M={M} N={N} K={K} mode={mode}: {ms*1e3:.1f} us")
