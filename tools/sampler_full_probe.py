"""Time the channel-0 sampler paths at B=32, V0=152697 (rocprofv3 --kernel-trace --stats around this script):
top-k 50 + top-p (the bench's setting: scan / collect / final kernels) against pure top-p and pure temperature
(no top_k: every row goes to the full-vocabulary path) and a huge top_k."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "moss-ttsd_amd"))
from mtts import capi
from mtts.engine import sampler_cfgs

lib = capi.lib()
rows, V = 32, 152697
rng = np.random.default_rng(0)
logits = torch.from_numpy((rng.standard_normal((rows, V)) * 2.5).astype(np.float32)).to(torch.bfloat16).cuda()
out = torch.zeros(rows, dtype=torch.int32, device="cuda")
for name, lc in (("topk50_topp", dict(top_k=50, top_p=0.95, temperature=1.0)), ("topp_only", dict(top_p=0.95)),
                 ("temperature_only", dict(temperature=1.1)), ("topk6000_topp", dict(top_k=6000, top_p=0.97))):
    cfg = sampler_cfgs([lc] * 8, [True] * 8)[0]
    ts = []
    for step in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        capi.check(lib.mtts_k_sample(logits.data_ptr(), rows, V, None, C.byref(cfg), -1, C.c_uint64(7), step, 0, out.data_ptr(), None))
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(name, "host ms per call (incl. scratch alloc):", [round(t * 1e3, 2) for t in ts], flush=True)
