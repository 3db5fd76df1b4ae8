"""Does the Infinity Cache (256 MB) serve a weight stream faster than HBM?  The decode gate/up GEMM (50 MB of weights, nt
loads) over 1 / 2 / 4 / 8 / 16 rotating weight copies (tuning aid, GPU box)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "moss-ttsd_amd"))
from mtts import capi
lib = capi.lib()
for name, N, K, epi in [("gate/up 50 MB", 12288, 2048, 2), ("down 25 MB", 2048, 6144, 0), ("qkv 16.8 MB", 4096, 2048, 0)]:
    res = []
    for copies in (1, 2, 4, 8, 16):
        us = C.c_float()
        capi.check(lib.mtts_k_gemm_bench(N, K, epi, 0, 8, copies, 64, C.byref(us)))
        res.append(f"{copies} copies: {us.value:.1f} us = {2.0 * N * K / us.value / 1e6:.2f} TB/s")
    print(name, " | ".join(res), flush=True)
