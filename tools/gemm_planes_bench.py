"""Time gemm_b3t_kernel alone (tuning aid, GPU box):  python tools/gemm_planes_bench.py [tile_code ...]
Shapes = the codec decoder's GEMMs at 8 windows per call; flags 0 none, 4 residual, 6 gamma+residual, 9 GELU+planes."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "moss-ttsd_amd"))
from mtts import capi
lib = capi.lib()
SHAPES = [("pw1", 24000, 4096, 512, 9), ("pw2", 24000, 512, 4096, 6), ("fc1", 12000, 3072, 768, 9), ("fc2", 12000, 768, 3072, 4),
          ("qkv", 12000, 2304, 768, 0), ("o", 12000, 768, 768, 4)]
only = os.environ.get("SHAPES")
codes = [int(x) for x in sys.argv[1:]] or [0]
for name, M, N, K, fl in SHAPES:
    if only and name not in only.split(","):
        continue
    for code in codes:
        us, used = C.c_float(), C.c_int32()
        rc = lib.mtts_k_gemm_planes_bench(M, N, K, fl, code, int(os.environ.get("ITERS", "20")), C.byref(us), C.byref(used))
        if rc:
            raise SystemExit(f"rc {rc}: {capi.codec_last_error() if hasattr(capi, 'codec_last_error') else ''}")
        tf = 3 * 2.0 * M * N * K / us.value / 1e6
        print(f"{name} M={M} N={N} K={K} flags={fl} tile={used.value}: {us.value:7.1f} us  {tf:6.0f} TFLOP/s (bf16x3 MFMA work)", flush=True)
