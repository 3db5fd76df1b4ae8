"""Time the REAL reference on this container's CPU cores (build container only: imports /root/reference; never runs on
the GPU box).  Writes profiles/r03_cpu_reference.json, which bench.py quotes as `cpu_baseline.reference_container`.

  * AR decode step: `AsteroidTTSInstruct.forward` (modeling_asteroid.py:337-426) at the ASSUMED 1.7B layer shape, bf16,
    B=32, KV length 4096, eager and SDPA attention, 2 and 4 decoder layers -> per-layer and fixed (embedding sum + 8
    heads) time -> the 28-layer step.  The KV cache is filled directly (random bf16 K/V through DynamicCache.update):
    prefilling 32 x 4096 tokens through the reference's all-position heads on a CPU is hours.
  * Codec: `XY_Tokenizer.decode` (XY_Tokenizer/xy_tokenizer/model.py:195-256) at full depth, one 375-code window.

Shims as in tests/golden/make_golden*.py (import-time only)."""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "moss-ttsd_amd"))


def time_ar(attn, layers, B, L, steps, threads):
    import make_golden as mg
    from mtts import synth
    from transformers.cache_utils import DynamicCache
    cfg = synth.make_config(num_hidden_layers=layers, max_position_embeddings=8192)
    hf = mg.ma.AsteroidTTSConfig(
        vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"],
        num_hidden_layers=layers, num_attention_heads=cfg["num_attention_heads"],
        num_key_value_heads=cfg["num_key_value_heads"], head_dim=cfg["head_dim"], max_position_embeddings=8192,
        rms_norm_eps=cfg["rms_norm_eps"], rope_theta=cfg["rope_theta"], tie_word_embeddings=True,
        speech_token_range=cfg["speech_token_range"], pad_token_id=cfg["pad_token_id"], eos_token_id=cfg["eos_token_id"],
        attn_implementation=attn, channels=8, speech_pad_token=1024, speech_vocab_size=1025)
    torch.manual_seed(0)
    m = mg.RefModel(hf).eval().to(torch.bfloat16)
    m.tie_weights()
    cache = DynamicCache(config=hf)
    for n in range(layers):
        k = torch.randn(B, cfg["num_key_value_heads"], L, 128).to(torch.bfloat16)
        v = torch.randn(B, cfg["num_key_value_heads"], L, 128).to(torch.bfloat16)
        cache.update(k, v, n)
    ids = torch.randint(0, 1024, (B, 1, 8))
    ids[..., 0] += 151665
    ts = []
    with torch.no_grad():
        for s in range(steps + 1):
            mask = torch.ones(B, L + s + 1, dtype=torch.long)
            pos = torch.full((B, 1), L + s, dtype=torch.long)
            t0 = time.perf_counter()
            out = m(input_ids=ids, attention_mask=mask, position_ids=pos, past_key_values=cache, use_cache=True, return_dict=True)
            _ = [l[:, -1].float() for l in out.logits_all]
            ts.append(time.perf_counter() - t0)
    del m, cache
    return min(ts[1:]), ts


def time_codec(threads):
    import importlib
    mgc = importlib.import_module("make_golden_codec")
    from mtts import synth_codec
    cfg = synth_codec.codec_config()
    w = synth_codec.synth_weights(cfg, 21)
    m = mgc.build_reference(cfg, w)
    codes = synth_codec.synth_codes(cfg, 22, [375])
    ts = []
    with torch.no_grad():
        for _ in range(2):
            t0 = time.perf_counter()
            m.decode([torch.from_numpy(c) for c in codes], overlap_seconds=10, device=torch.device("cpu"))
            ts.append(time.perf_counter() - t0)
    return min(ts)


if __name__ == "__main__":
    threads = os.cpu_count()
    torch.set_num_threads(threads)
    B, L = 32, 4096
    rec = {"cores": threads, "torch": torch.__version__, "transformers": __import__("transformers").__version__,
           "dtype": "bf16", "batch": B, "kv_len": L, "what": "reference AsteroidTTSInstruct.forward, one decode step; "
           "KV cache filled directly; 28-layer step = fixed + 28 x per-layer from the 2- and 4-layer timings"}
    for attn in ("eager", "sdpa"):
        t2, all2 = time_ar(attn, 2, B, L, 3, threads)
        t4, all4 = time_ar(attn, 4, B, L, 3, threads)
        per_layer = (t4 - t2) / 2
        fixed = t2 - 2 * per_layer
        step28 = fixed + 28 * per_layer
        rec[attn] = {"s_per_step_2_layers": t2, "s_per_step_4_layers": t4, "s_per_layer": per_layer, "s_fixed": fixed,
                     "s_per_step_28_layers": step28, "codec_ids_per_s": 8 * B / step28, "frames_per_s": B / step28,
                     "all_2": all2, "all_4": all4}
        print(attn, rec[attn], flush=True)
    tc = time_codec(threads)
    rec["codec_decode"] = {"s_per_375_code_window": tc, "audio_s_per_s": 30.0 / tc, "what": "XY_Tokenizer.decode, full depth, fp32"}
    print(rec["codec_decode"])
    with open(os.path.join(ROOT, "profiles", "r03_cpu_reference.json"), "w") as f:
        json.dump(rec, f, indent=1)
