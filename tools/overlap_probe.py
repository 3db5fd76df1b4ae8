"""How well do the HBM-bound decode loop and the MFMA-bound codec share the GPU?  (tuning aid, GPU box only)
Runs 64 decode steps at B=32 / L~4 k on the default stream while the codec decodes 8-window batches on a side
stream that is (a) a plain stream, (b)... restricted to a CU mask via hipExtStreamCreateWithCUMask."""
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "moss-ttsd_amd"))
sys.path.insert(0, ROOT)
from bench import make_weights_on_device  # noqa: E402
from mtts import capi, synth, synth_codec  # noqa: E402
from mtts.codec import CodecEngine  # noqa: E402
from mtts.engine import Engine  # noqa: E402

dev = torch.device("cuda:0")
hip = C.CDLL("libamdhip64.so")


def masked_stream(words):
    arr = (C.c_uint32 * len(words))(*words)
    h = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), C.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value, device=dev)


cfg = synth.assumed_1p7b()
B, L, T = 32, 4096, 512
eng = Engine(cfg, max_batch=B, max_seq_len=L + 64, device=str(dev))
for name, t in make_weights_on_device(cfg, 1234, dev, 0, 1):
    eng.bind(name, t)
    del t
capi.check(eng.lib.mtts_weights_ready(eng._h))
ids, mask = synth.synth_prompts(cfg, 77, B, T, audio_frac=0.5, ragged=False)
layers = [dict(top_k=50, top_p=0.95, temperature=1.0, repetition_penalty=1.0)] * 8
eng.begin(ids, mask, T + (L - (T - 7)) + 8, layers=layers, do_samples=[True] * 8, seed=1)
eng.debug_set_kv_len(L - 1100)
ccfg = synth_codec.codec_config()
cod = CodecEngine(ccfg, device=str(dev))
cod.bind_state_dict(synth_codec.synth_weights(ccfg, 5))
codes = torch.randint(0, 1024, (8, 8, 375), device=dev)
cod.detokenize(codes, [375] * 8)
eng.step(8)
eng.sync_state()
torch.cuda.synchronize()


DEC = None          # stream the decode loop runs on (None = default stream)


def decode_alone(n=64):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.step(n, DEC)
    eng.sync_state(DEC)
    return (time.perf_counter() - t0) / n * 1e3


def codec_alone(stream, reps=4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        cod.detokenize_async(codes, [375] * 8, stream)
    stream.synchronize()
    return (time.perf_counter() - t0) / (reps * 8) * 1e3


def both(stream, n=64, reps=4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    keep = [cod.detokenize_async(codes, [375] * 8, stream) for _ in range(reps)]
    eng.step(n, DEC)
    eng.sync_state(DEC)
    t_dec = time.perf_counter() - t0
    stream.synchronize()
    t_all = time.perf_counter() - t0
    return t_dec / n * 1e3, t_all * 1e3


out = {}
FULL = 0xFFFFFFFF
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
decs = {"default stream": None, "own stream": torch.cuda.Stream(device=dev), "own high-priority stream": torch.cuda.Stream(device=dev, priority=-1)}
masks = {
    "plain": None,
    "plain low-priority": "low",
    "cu mask low 128": [FULL] * 4 + [0] * 4,
    "cu mask low 192": [FULL] * 6 + [0] * 2,
    "cu mask low 64": [FULL] * 2 + [0] * 6,
}
for dname, dstream in decs.items():
    DEC = dstream
    alone = decode_alone()
    for name, m in masks.items():
        key = f"decode on {dname} / codec on {name}"
        try:
            if m is None:
                st = torch.cuda.Stream(device=dev)
            elif m == "low":
                st = torch.cuda.Stream(device=dev, priority=0)
            else:
                st = masked_stream(m)
            ca = codec_alone(st)
            d, tot = both(st)
            seq = 64 * alone + 4 * 8 * ca
            out[key] = {"decode_alone": round(alone, 3), "codec_alone_ms_per_window": round(ca, 2),
                        "decode_ms_per_step_beside_codec": round(d, 3), "both_wall_ms": round(tot, 1),
                        "sequential_would_be_ms": round(seq, 1)}
        except Exception as e:  # noqa: BLE001
            out[key] = "failed: %r" % (e,)
        print(key, out[key], flush=True)
print(json.dumps(out))
