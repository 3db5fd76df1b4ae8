"""Do two codec engines on two HIP streams fill each other's tile-quantisation tails?  (tuning aid, GPU box only)
8 windows through one engine in one call vs 4 + 4 windows through two engines on two streams."""
import json
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "moss-ttsd_amd"))
sys.path.insert(0, ROOT)
from mtts import synth_codec  # noqa: E402
from mtts.codec import CodecEngine  # noqa: E402

dev = torch.device("cuda:0")
cfg = synth_codec.codec_config()
w = synth_codec.synth_weights(cfg, 5)
engs = [CodecEngine(cfg, device=str(dev)) for _ in range(2)]
for e in engs:
    e.bind_state_dict(w)
T = 375
codes8 = torch.randint(0, 1024, (cfg["nq"], 8, T), device=dev)
halves = [codes8[:, :4].contiguous(), codes8[:, 4:].contiguous()]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
engs[0].detokenize(codes8, [T] * 8)
for e, h in zip(engs, halves):
    e.detokenize(h, [T] * 4)
torch.cuda.synchronize()


def one(reps=3):
    t0 = time.perf_counter()
    for _ in range(reps):
        engs[0].detokenize(codes8, [T] * 8)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / 8 * 1e3


def two(reps=3):
    t0 = time.perf_counter()
    for _ in range(reps):
        outs = [e.detokenize_async(h, [T] * 4, s) for e, h, s in zip(engs, halves, streams)]
        for e, s in zip(engs, streams):
            e.check(s)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / 8 * 1e3


def serial4(reps=3):
    t0 = time.perf_counter()
    for _ in range(reps):
        for e, h in zip(engs, halves):
            e.detokenize(h, [T] * 4)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / 8 * 1e3


print(json.dumps({"one_engine_8_windows_ms_per_window": one(), "two_engines_4+4_two_streams": two(),
                  "two_engines_4+4_serial": serial4(), "again_one": one(), "again_two": two()}))
