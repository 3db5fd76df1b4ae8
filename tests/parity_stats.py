"""Agreement of the HIP engine with the reference's greedy runs over ALL decisions (no margin gate), per fixture.
and the codec decoder's waveform error against the reference's fixtures.  GPU box only.
Writes gpurun_out/r01_parity_stats.json and gpurun_out/r01_codec_parity_stats.json (copied to profiles/)."""
import json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "moss-ttsd_amd")); sys.path.insert(0, ROOT)
import numpy as np
from mtts import synth
from mtts.engine import Engine
from oracle import asteroid_oracle as ao

out = {}
for name in ["ar_text_ragged", "ar_flush0", "ar_audio_tail", "ar_gqa4", "ar_rep_penalty"]:
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    cfg = json.loads(str(z["cfg"]))
    w = synth.synth_weights(cfg, int(z["seed"]), **json.loads(str(z["wkw"])))
    eng = Engine(cfg, max_batch=4, max_seq_len=256)
    eng.bind_state_dict(w)
    gold = z["out_ids"]; T = z["input_ids"].shape[1]
    layers = json.loads(str(z["layers"])) or None
    _, dec = eng.generate(z["input_ids"], z["attention_mask"], int(z["max_length"]), layers=layers, forced=gold)
    orc = ao.AsteroidOracle(cfg, w, "bf16")
    _, odec, _ = orc.generate(z["input_ids"], z["attention_mask"], int(z["max_length"]), layers=layers, forced=gold)
    want = gold[:, T - 7:].transpose(1, 0, 2)
    m = z["margins"]
    used = m < 9.0
    low = used & (m < 0.02)
    out[name] = {
        "decisions": int(used.sum()), "low_margin_decisions": int(low.sum()),
        "hip_equals_reference": int((dec[used] == want[used]).sum()),
        "hip_equals_reference_low_margin": int((dec[low] == want[low]).sum()),
        "oracle_equals_reference": int((odec[used] == want[used]).sum()),
        "hip_equals_oracle": int((dec[used] == odec[used]).sum()),
        "exact_ties_in_reference": int((used & (m == 0)).sum()),
    }
    eng.close()
    print(name, out[name], flush=True)
tot = {k: sum(v[k] for v in out.values()) for k in next(iter(out.values()))}
out["total"] = tot
print("total", tot)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r01_parity_stats.json"), "w"), indent=1)

# ---- codec decoder: waveform RMS error against the reference's fixtures, both GEMM modes ------------------------
import torch
from mtts import synth_codec
from mtts.codec import CodecEngine

codec = {}
for mode in ("bf16x3", "f32"):
    os.environ["MTTS_CODEC_GEMM"] = mode
    for name in ["codec_T40", "codec_ragged_1win", "codec_T600", "codec_full_T24"]:
        z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        cfg = json.loads(str(z["cfg"]))
        w = synth_codec.synth_weights(cfg, int(z["seed"]))
        codes = synth_codec.synth_codes(cfg, int(z["seed"]) + 1, list(z["lengths"]))
        eng = CodecEngine(cfg)
        eng.bind_state_dict(w)
        wavs = [x.cpu().numpy() for x in eng.decode([torch.from_numpy(c) for c in codes])]
        eng.close()
        stride = int(z["stride"])
        errs, sig = [], []
        for i, wv in enumerate(wavs):
            ref = z[f"wav{i}_sub"].astype(np.float64)
            errs.append(float(np.sqrt(np.mean((wv[::stride].astype(np.float64) - ref) ** 2))))
            sig.append(float(np.sqrt(np.mean(ref ** 2))))
        codec.setdefault(name, {})[mode] = {"max_rms_error": max(errs), "signal_rms": max(sig)}
        print(name, mode, codec[name][mode])
json.dump(codec, open(os.path.join(ROOT, "gpurun_out", "r01_codec_parity_stats.json"), "w"), indent=1)
