"""Time the codec GEMM on its dominant shapes, exact f32 MFMA and bf16x3 (tuning aid, GPU box only)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "moss-ttsd_amd"))
import torch
from mtts import capi, codec
lib = capi.lib()
for (M, N, K) in [(24000, 4096, 512), (24000, 512, 4096), (12000, 3072, 768), (12000, 768, 3072), (12000, 2304, 768), (24000, 960, 976)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda"); c = torch.empty(M, N, device="cuda")
    ref = None
    for mode, act in (("f32", 1), ("bf16x3", 1 | 0x100)):
        for _ in range(2):
            lib.mtts_k_gemm_f32(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, act, None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.mtts_k_gemm_f32(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, act, None)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        err = "" if ref is None else f"  max |diff| vs f32 {float((c - ref).abs().max()):.2e} (|c| max {float(ref.abs().max()):.2f})"
        if ref is None:
            ref = c.clone()
        print(f"M={M} N={N} K={K} {mode}: {ms*1e3:.0f} us  {2*M*N*K/ms/1e9:.1f} TFLOP/s{err}", flush=True)
