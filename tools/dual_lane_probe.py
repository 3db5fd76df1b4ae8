"""Do two half-batch decode chains overlap on one MI355X?  (tuning aid, GPU box only)

A decode step is a chain of ~290 dependent kernels; the attention passes stream HBM at copy rate, everything else is
mostly launch/latency bubbles.  Two independent half-batches on two HIP streams could fill each other's bubbles at the
price of streaming the weights twice.  This probe measures it with what exists: two Engine objects (own weights each),
16 dialogues each at a ~4 k context, stepped (a) one after the other, (b) concurrently on two streams, against (c) one
engine with all 32 dialogues."""
import json
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "moss-ttsd_amd"))
sys.path.insert(0, ROOT)
from bench import make_weights_on_device  # noqa: E402
from mtts import capi, synth  # noqa: E402
from mtts.engine import Engine  # noqa: E402

dev = torch.device("cuda:0")
cfg = synth.assumed_1p7b()
L, T = int(os.environ.get("CTX", 4096)), 512
layers = [dict(top_k=50, top_p=0.95, temperature=1.0, repetition_penalty=1.0)] * 8


def mk(B, seed):
    eng = Engine(cfg, max_batch=B, max_seq_len=L + 64, device=str(dev))
    for name, t in make_weights_on_device(cfg, 1234, dev, 0, 1):
        eng.bind(name, t)
        del t
    capi.check(eng.lib.mtts_weights_ready(eng._h))
    ids, mask = synth.synth_prompts(cfg, seed, B, T, audio_frac=0.5, ragged=False)
    eng.begin(ids, mask, T + (L - (T - 7)) + 8, layers=layers, do_samples=[True] * 8, seed=seed)
    eng.debug_set_kv_len(L - 400)
    eng.step(4)
    eng.sync_state()
    return eng


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


out = {}
N = 48
for total in (32, 8, 2):
    half = total // 2
    one = mk(total, 5)
    out[f"one_engine_B{total}_ms"] = timed(lambda n: (one.step(n), one.sync_state()), N)
    one.close()
    a, b = mk(half, 6), mk(half, 7)
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

    def serial(n):
        a.step(n, sa); a.sync_state(sa)
        b.step(n, sb); b.sync_state(sb)

    def concurrent(n):
        for _ in range(n):            # interleave the submissions so that both queues are fed
            a.step(1, sa)
            b.step(1, sb)
        a.sync_state(sa); b.sync_state(sb)

    out[f"two_engines_B{half}_serial_ms"] = timed(serial, N)
    out[f"two_engines_B{half}_concurrent_ms"] = timed(concurrent, N)
    a.close(); b.close()
    print(json.dumps(out), flush=True)
