"""Agreement of the HIP engine with the reference's greedy runs over ALL decisions (no margin gate), per fixture.
and the codec decoder's waveform error against the reference's fixtures.  GPU box only.
Writes gpurun_out/r02_parity_stats.json and gpurun_out/r02_codec_parity_stats.json (copied to profiles/).
Every decision where the engine differs from the reference is listed with the two candidates' logits as the engine and
as the oracle (which agrees with the reference there or not) computed them."""
import json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "moss-ttsd_amd")); sys.path.insert(0, ROOT)
import numpy as np
from mtts import synth
from mtts.engine import Engine
from oracle import asteroid_oracle as ao

out = {}
for name in ["ar_text_ragged", "ar_flush0", "ar_audio_tail", "ar_gqa4", "ar_rep_penalty", "ar_flush_past_max"]:
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
    # anatomy of every miss: replay to that step with the step API and read the engine's logits of the two candidates
    misses = np.argwhere(used & (dec != want))
    out[name]["misses"] = []
    if len(misses):
        _, _, ologs = orc.generate(z["input_ids"], z["attention_mask"], int(z["max_length"]), layers=layers, forced=gold, return_logits=True)
        for (s_, b_, c_) in misses:
            # the logits that decide step s_ are those left by the forward of step s_-1: replay exactly s_ forced steps
            if int(s_) == 0:
                eng.begin(z["input_ids"], z["attention_mask"], int(z["max_length"]), layers=layers)
            else:
                eng.generate(z["input_ids"], z["attention_mask"], int(z["max_length"]), layers=layers, forced=gold[:, :T - 7 + int(s_)])
            l0, l17 = eng.read_logits()
            hl = (l0 if c_ == 0 else l17[c_ - 1])[b_]
            ol = ologs[int(s_)][int(c_)][b_]
            a, h = int(want[s_, b_, c_]), int(dec[s_, b_, c_])
            out[name]["misses"].append({"step": int(s_), "row": int(b_), "channel": int(c_), "reference_token": a, "hip_token": h,
                                        "reference_margin": float(m[s_, b_, c_]),
                                        "hip_logits(ref_tok,hip_tok)": [float(hl[a]), float(hl[h])],
                                        "oracle_logits(ref_tok,hip_tok)": [float(ol[a]), float(ol[h])],
                                        "oracle_token": int(odec[s_, b_, c_])})
    eng.close()
    print(name, out[name], flush=True)
tot = {k: sum(v[k] for v in out.values()) for k in next(iter(out.values())) if k != "misses"}
out["total"] = tot
print("total", tot)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r02_parity_stats.json"), "w"), indent=1)

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
json.dump(codec, open(os.path.join(ROOT, "gpurun_out", "r02_codec_parity_stats.json"), "w"), indent=1)
