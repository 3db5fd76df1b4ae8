"""SURVEY.md §8d config 5 end to end on one GPU (synthetic weights of the ASSUMED dims, GPU box only):
B = 8 dialogues, each with a 20 s / 16 kHz voice-clone prompt -> XY_Tokenizer encode -> prompt = 128 text tokens +
the 250 prompt frames -> prefill -> decode to an 8192-token KV context -> XY_Tokenizer decode of every frame.
Prints one JSON line with the wall time of each stage."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "moss-ttsd_amd"))
sys.path.insert(0, ROOT)
from bench import make_weights_on_device  # noqa: E402
from mtts import capi, synth, synth_codec  # noqa: E402
from mtts.codec import CodecEngine  # noqa: E402
from mtts.engine import Engine  # noqa: E402

B, L, TEXT = 8, int(os.environ.get("CONFIG5_CONTEXT", 8192)), 128
dev = torch.device("cuda:0")
cfg = synth.assumed_1p7b()
ccfg = synth_codec.codec_config()
out = {"workload": f"config 5: batch {B}, 20 s voice-clone prompt each, decode to a {L}-token context, codec both ways"}

t0 = time.perf_counter()
cod = CodecEngine(ccfg, device=str(dev))
cod.bind_state_dict(synth_codec.synth_weights(ccfg, 5, encoder=True))
eng = Engine(cfg, max_batch=B, max_seq_len=L + 64, device=str(dev))
for name, t in make_weights_on_device(cfg, 1234, dev, 0, 1):
    eng.bind(name, t)
    del t
capi.check(eng.lib.mtts_weights_ready(eng._h))
torch.cuda.synchronize()
out["setup_s"] = time.perf_counter() - t0

wavs = synth_codec.synth_wavs(3, [20 * 16000] * B)
cod.encode(wavs[:1])                                   # warm-up (workspace)
torch.cuda.synchronize()
t0 = time.perf_counter()
codes = cod.encode(wavs)                               # list of [8, 250] int64 (device)
torch.cuda.synchronize()
out["encode_s"] = time.perf_counter() - t0
out["prompt_frames"] = [int(c.shape[-1]) for c in codes]

rng = np.random.default_rng(11)
seqs = []
for b in range(B):
    a = codes[b].cpu().numpy().T                       # [250, 8]
    raw = np.full((TEXT + a.shape[0], 8), synth.SPEECH_PAD, dtype=np.int64)
    raw[:TEXT, 0] = rng.integers(0, 151643, TEXT)
    raw[TEXT:, 0] = synth.SPEECH_OFFSET + a[:, 0]
    raw[TEXT:, 1:] = a[:, 1:]
    seqs.append(synth.shifting_inputs(raw, cfg["pad_token_id"]))
ids, mask = synth.left_pad(seqs, cfg["pad_token_id"])
T = ids.shape[1]
n_real = T - 7
max_length = T + (L - n_real) + 8
layers = [dict(top_k=50, top_p=0.95, temperature=1.0, repetition_penalty=1.0)] * 8

torch.cuda.synchronize()
t0 = time.perf_counter()
eng.begin(ids, mask, max_length, layers=layers, do_samples=[True] * 8, seed=7)
eng.sync_state()
out["prefill_s"] = time.perf_counter() - t0
out["prompt_tokens"] = int(B * n_real)

steps = L - n_real
t0 = time.perf_counter()
done = 0
marks = []
while done < steps:
    n = min(256, steps - done)
    eng.step(n)
    done += n
    st, fin = eng.sync_state()
    marks.append((done, time.perf_counter() - t0))
    assert not fin, "a dialogue finished early (synthetic weights keep channel 0 in the speech range)"
out["decode_s"] = time.perf_counter() - t0
out["decode_steps"] = steps
out["ms_per_step_mean"] = out["decode_s"] / steps * 1e3
(d0, t_a), (d1, t_b) = marks[-3], marks[-1]
out["ms_per_step_at_full_context"] = (t_b - t_a) / (d1 - d0) * 1e3

gen = eng.read_generated(steps)                         # [G,B,8]
n = gen.shape[0] - 7
cc = np.stack([gen[j:n + j, :, j] for j in range(8)], axis=0)
cc[0] -= synth.SPEECH_OFFSET
cc = np.clip(cc, 0, 1023).transpose(0, 2, 1)            # [8,B,n]
tc = torch.from_numpy(np.ascontiguousarray(cc)).to(dev)
cod.detokenize(tc[:, :1, :375].contiguous(), [375])
torch.cuda.synchronize()
t0 = time.perf_counter()
wv = cod.decode([tc[:, b] for b in range(B)])
torch.cuda.synchronize()
out["codec_decode_s"] = time.perf_counter() - t0
audio_s = sum(int(w.shape[0]) for w in wv) / 24000.0
wall = out["encode_s"] + out["prefill_s"] + out["decode_s"] + out["codec_decode_s"]
out["audio_seconds"] = audio_s
out["wall_s"] = wall
out["real_time_factor"] = audio_s / wall
out["codec_ids_per_s"] = B * n * 8 / wall
print(json.dumps(out))
