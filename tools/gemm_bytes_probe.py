"""What would 13/16 of the weight bytes buy the skinny GEMMs of a decode step?  Same launches with K cut to 13/16
(fewer bytes, same tiles): an upper bound for a lossless 13-bit weight format (sizing aid, GPU box only)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "moss-ttsd_amd"))
from mtts import capi
lib = capi.lib()
shapes = [("qkv", 4096, 2048, 0), ("o", 2048, 2048, 0), ("gateup", 12288, 2048, 2), ("down", 2048, 6144, 0), ("head0", 152704, 2048, 1)]
tot = [0.0, 0.0]
for name, N, K, epi in shapes:
    row = []
    for i, k in enumerate((K, K * 13 // 16)):
        mb = N * k * 2 / 1e6
        copies = max(2, int(600 / mb) + 1)
        best = 1e9
        for ks in ([1] if epi else [1, 2, 4, 8]):
            for wv in (2, 4, 8):
                if (k // 16) // ks // wv < 1:
                    continue
                us = C.c_float()
                if lib.mtts_k_gemm_bench(N, k, epi, ks, wv, copies, copies * 3, C.byref(us)) == 0:
                    best = min(best, us.value)
        row.append((k, mb, best))
        tot[i] += best * (1 if name == "head0" else 28)
    print(name, " | ".join(f"K={k}: {mb:.1f} MB {u:.2f} us" for k, mb, u in row), flush=True)
print("per step (28 layers + head0): %.1f us -> %.1f us" % (tot[0], tot[1]))
