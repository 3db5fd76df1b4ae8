"""Is the gate/up GEMM (384 N-tiles on 256 CUs) limited by per-CU load imbalance?  Times the skinny GEMM at
N = 256 / 384 / 512 / 768 tiles of 32 columns (K = 2048, SwiGLU epilogue) and the heads (tuning aid, GPU box only)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "moss-ttsd_amd"))
from mtts import capi
lib = capi.lib()
for name, N, K, epi, ks, wv in [("gu256", 8192, 2048, 2, 1, 8), ("gu384", 12288, 2048, 2, 1, 8), ("gu512", 16384, 2048, 2, 1, 8),
                                ("gu768", 24576, 2048, 2, 1, 8), ("gu384w4", 12288, 2048, 2, 1, 4),
                                ("head0w4", 152704, 2048, 1, 1, 4), ("head0w8", 152704, 2048, 1, 1, 8), ("head0w2", 152704, 2048, 1, 1, 2),
                                ("qkv", 4096, 2048, 0, 2, 8), ("o", 2048, 2048, 0, 4, 8), ("down", 2048, 6144, 0, 4, 8)]:
    mb = N * K * 2 / 1e6
    copies = max(2, int(600 / mb) + 1)
    us = C.c_float()
    rc = lib.mtts_k_gemm_bench(N, K, epi, ks, wv, copies, copies * 4, C.byref(us))
    print(name, f"{mb:.1f} MB", "error" if rc else f"{us.value:.2f} us {mb / us.value * 1e3:.0f} GB/s", flush=True)
