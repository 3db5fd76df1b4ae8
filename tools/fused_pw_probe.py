"""Fused Vocos pw1 -> GELU -> pw2 kernel (MTTS_CODEC_FUSED_PW) against the two-launch form: waveform difference on a
full-depth window and the decoder's time per window at 8 and 32 windows per call."""
import json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "moss-ttsd_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from mtts import synth_codec
from mtts.codec import CodecEngine

cfg = synth_codec.codec_config()
w = synth_codec.synth_weights(cfg, 21)
z = np.load(os.path.join(ROOT, "tests", "golden", "codec_full_T375.npz"))
codes = synth_codec.synth_codes(cfg, 22, [375])
out = {}
wav = {}
for mode in ("0", os.environ.get("PROBE_MODE", "1")):
    if mode == "auto":
        os.environ.pop("MTTS_CODEC_FUSED_PW", None)
    else:
        os.environ["MTTS_CODEC_FUSED_PW"] = mode
    eng = CodecEngine(cfg)
    eng.bind_state_dict(w)
    y = eng.decode([torch.from_numpy(c) for c in codes])[0].cpu().numpy()
    wav[mode] = y
    ref = z["wav0_sub"].astype(np.float64)
    out["rms_vs_reference_fused=" + mode] = float(np.sqrt(np.mean((y[::int(z["stride"])].astype(np.float64) - ref) ** 2)))
    for windows in [int(x) for x in os.environ.get("PROBE_WINDOWS", "8,32").split(",")]:
        c = torch.randint(0, 1024, (cfg["nq"], windows, 375), device="cuda")
        eng.detokenize(c, [375] * windows)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.detokenize(c, [375] * windows)
        torch.cuda.synchronize()
        out[f"ms_per_window_fused={mode}_at_{windows}"] = (time.perf_counter() - t0) / 3 / windows * 1e3
    eng.close()
out["rms_fused_vs_two_launch"] = float(np.sqrt(np.mean((wav["0"].astype(np.float64) - wav[os.environ.get("PROBE_MODE", "1")]) ** 2)))
print(json.dumps(out, indent=1))
