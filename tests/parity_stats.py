"""Agreement of the HIP engine with the reference's greedy runs over ALL decisions (no margin gate), per fixture.
GPU box only.  Writes profiles/r01_parity_stats.json."""
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
