"""Time the tiled prefill GEMM (512 rows) on the decoder-layer shapes (tuning aid, GPU box only)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "moss-ttsd_amd"))
from mtts import capi
lib = capi.lib()
R = int(os.environ.get("TILE_ROWS", 512))
for name, N, K, epi, kss in [("qkv", 4096, 2048, 0, (1, 2, 4)), ("o", 2048, 2048, 0, (1, 2, 4, 8)), ("gateup", 12288, 2048, 2, (1,)),
                             ("down", 2048, 6144, 0, (1, 2, 4, 8))]:
    res = []
    for ks in kss:
        us = C.c_float()
        capi.check(lib.mtts_k_gemm_bench(N, K, epi, ks, -R, 4, 40, C.byref(us)))
        res.append(f"ks{ks}: {us.value:.1f}us {2.0 * R * N * K / us.value / 1e6:.0f} TFLOP/s")
    print(name, " | ".join(res), flush=True)
