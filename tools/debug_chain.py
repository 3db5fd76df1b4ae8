import sys, os, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT,'moss-ttsd_amd')); sys.path.insert(0, ROOT)
from mtts import synth
from mtts.engine import Engine
cfg = synth.tiny()
w = synth.synth_weights(cfg, 231, emb_row_sigma=0.6, speech_boost=6.0, eos_boost=1.0)
ids, mask = synth.synth_prompts(cfg, 232, 3, 24, 0.0, False)
T = ids.shape[1]; base=T-7; M=12; max_length=base+M; G=M+26
rng = np.random.default_rng(3)
forced = np.zeros((3, base + G, 8), dtype=np.int64)
forced[:, :base] = ids[:, :base]
forced[:, base:, 0] = 151665 + rng.integers(0, 1024, (3, G))
forced[:, base:, 1:] = rng.integers(0, 1024, (3, G, 7))
forced[0, base + M - 2, 0] = 77
forced[1, base + M + 3, 0] = 77
forced[2, base + M + 8, 0] = 77
eng = Engine(cfg, max_batch=4, max_seq_len=256)
eng.bind_state_dict(w)
out, dec = eng.generate(ids, mask, max_length, forced=forced, forced_as_draw="all")
print(out.shape)
print(out[:, base:, 0])
print("dec ch0", dec[:, :, 0].T)
print(eng.seq_state())
