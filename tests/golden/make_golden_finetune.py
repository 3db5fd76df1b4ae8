"""Fine-tune data path fixtures from the REFERENCE's own `process_inputs`
(/root/reference/finetune/data_preprocess.py:26-147).  Build container only.

The reference function is called with a stub tokenizer (code points) and a stub `spt` whose `encode` returns fixed codes per
waveform length, so only the function's own work is pinned: segment layout, the 151665 offset on channel 0, channel
alignment, the -100 label masks, the learned <|end_of_speech|> label, both audio formats and the length bookkeeping.
Only data is written (tests/golden/finetune_inputs.json)."""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
import transformers  # noqa: F401
for n in ["liger_kernel", "liger_kernel.transformers", "liger_kernel.transformers.model",
          "liger_kernel.transformers.model.loss_utils", "torchaudio", "torchaudio.functional",
          "torchaudio.functional.functional", "librosa"]:
    sys.modules[n] = types.ModuleType(n)
sys.modules["liger_kernel.transformers.model.loss_utils"].LigerForCausalLMLoss = None
sys.modules["torchaudio.functional.functional"]._hz_to_mel = None
sys.modules["torchaudio.functional.functional"]._mel_to_hz = None
sys.modules["torchaudio"].functional = sys.modules["torchaudio.functional"]
sys.modules["torchaudio.functional"].functional = sys.modules["torchaudio.functional.functional"]
sys.path.insert(0, "/root/reference")
sys.path.insert(0, "/root/reference/finetune")
import data_preprocess as ref  # noqa: E402


class StubTokenizer:
    """encode = code points (+3 so that no id is 0); a leading BOS id 1 when add_special_tokens."""
    pad_token_id = 151643

    def encode(self, s, add_special_tokens=True):
        ids = [min(ord(c) + 3, 151000) for c in s]
        return ([1] + ids) if add_special_tokens and os.environ.get("STUB_BOS") else ids


class StubSpt:
    """encode([wav]) -> codes (nq, len // 1280) taken from a table filled by the caller."""
    def __init__(self, table):
        self.table = table

    def encode(self, wav_list, **_):
        return {"codes_list": [torch.from_numpy(self.table[int(w.shape[-1])]) for w in wav_list]}


def run(case):
    rng = np.random.default_rng(case["seed"])
    table, kw = {}, {}
    for key in ("audio_data", "reference_audio", "main_audio"):
        n = case.get(key)
        if n is None:
            continue
        nq = case.get("nq", 8)
        table[n] = rng.integers(0, 1024, (nq, n // 1280)).astype(np.int64)
        wav = torch.from_numpy(rng.standard_normal(n).astype(np.float32))
        kw[key] = wav if case.get("flat") else wav[None]
    codes = {str(k): v.tolist() for k, v in table.items()}      # (the reference adds its offset IN PLACE, through the numpy view)
    ids, labels, total, audio = ref.process_inputs(StubTokenizer(), StubSpt(table), case["prompt"], case["text"], "cpu", **kw)
    return {"case": case, "codes": codes, "input_ids": np.asarray(ids).tolist(),
            "labels": np.asarray(labels).tolist(), "total_length": int(total), "audio_length": int(audio)}


if __name__ == "__main__":
    cases = [
        dict(seed=1, prompt="You are a synthesizer.", text="<speaker1>hello<speaker2>world", audio_data=16000 * 2),
        dict(seed=2, prompt="p", text="[S1]你好。[S2]再见", reference_audio=12800, main_audio=16000 * 3 + 7),
        dict(seed=3, prompt="", text="x", reference_audio=2560, main_audio=1280, flat=True),     # 1-D waveforms
        dict(seed=4, prompt="style", text="fewer channels", audio_data=12800, nq=4),             # channel alignment: pad
        dict(seed=5, prompt="style", text="more channels", audio_data=6400, nq=10),              # ... and cut
    ]
    out = [run(c) for c in cases]
    err = None
    try:
        ref.process_inputs(StubTokenizer(), StubSpt({}), "p", "t", "cpu")
    except ValueError as e:
        err = str(e)
    json.dump({"cases": out, "no_audio_error": err}, open(os.path.join(HERE, "finetune_inputs.json"), "w"), ensure_ascii=False)
    for o in out:
        print(o["case"], np.asarray(o["input_ids"]).shape, o["total_length"], o["audio_length"])
    print("no audio ->", err)
