"""Sweep (ksplit, waves) of the skinny GEMM on the decode shapes (tuning aid, GPU box only)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "moss-ttsd_amd"))
from mtts import capi
lib = capi.lib()
shapes = [("qkv", 4096, 2048, 0), ("o", 2048, 2048, 0), ("gateup", 12288, 2048, 2), ("down", 2048, 6144, 0), ("head0", 152704, 2048, 1)]
for name, N, K, epi in shapes:
    mb = N * K * 2 / 1e6
    copies = max(2, int(600 / mb) + 1)
    res = []
    for ks in ([1] if epi else [1, 2, 4, 8]):
        for wv in (2, 4, 8):
            if (K // 16) // ks // wv < 1:
                continue
            us = C.c_float()
            rc = lib.mtts_k_gemm_bench(N, K, epi, ks, wv, copies, copies * 3, C.byref(us))
            if rc:
                print(name, ks, wv, "error", lib.mtts_last_error())
                continue
            res.append((us.value, ks, wv))
    res.sort()
    print(name, f"{mb:.1f} MB", " | ".join(f"ks{k} w{w}: {u:.1f}us {mb / u * 1e-3 * 1e3:.0f}GB/s" for u, k, w in (res if "--all" in sys.argv else res[:6])), flush=True)
