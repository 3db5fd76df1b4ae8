"""Fine-tune data path against the reference's own `process_inputs`
(tests/golden/finetune_inputs.json, written by tests/golden/make_golden_finetune.py from
/root/reference/finetune/data_preprocess.py:26-147 with a stub tokenizer and a stub spt).  CPU only."""
import json
import os
import pickle

import numpy as np
import pytest
import torch


class StubTokenizer:          # the same stub the fixture script used
    pad_token_id = 151643

    def encode(self, s, add_special_tokens=True):
        return [min(ord(c) + 3, 151000) for c in s]


class StubSpt:
    output_sample_rate = 24000

    def __init__(self, table):
        self.table = table
        self.calls = []

    def to(self, device):
        return self

    def encode(self, wav_list, **_):
        self.calls.append([int(w.shape[-1]) for w in wav_list])
        return {"codes_list": [torch.from_numpy(self.table[int(w.shape[-1])]) for w in wav_list]}


@pytest.fixture(scope="module")
def dp():
    from finetune import data_preprocess
    return data_preprocess


def test_process_inputs_matches_reference(golden_dir, dp):
    fx = json.load(open(os.path.join(golden_dir, "finetune_inputs.json")))
    assert len(fx["cases"]) >= 5
    for rec in fx["cases"]:
        case = rec["case"]
        table = {int(k): np.array(v, dtype=np.int64) for k, v in rec["codes"].items()}
        rng = np.random.default_rng(case["seed"])
        kw = {}
        for key in ("audio_data", "reference_audio", "main_audio"):
            n = case.get(key)
            if n is None:
                continue
            rng.integers(0, 1024, (case.get("nq", 8), n // 1280))          # keep the generator in step with the script
            wav = torch.from_numpy(rng.standard_normal(n).astype(np.float32))
            kw[key] = wav if case.get("flat") else wav[None]
        spt = StubSpt(table)
        ids, labels, total, audio = dp.process_inputs(StubTokenizer(), spt, case["prompt"], case["text"], "cpu", **kw)
        assert np.array_equal(ids, np.array(rec["input_ids"])), case
        assert np.array_equal(labels, np.array(rec["labels"])), case
        assert (total, audio) == (rec["total_length"], rec["audio_length"])
        # two recordings are encoded in two calls (reference: :68-69), one recording in one
        assert len(spt.calls) == (2 if "reference_audio" in case else 1)
        # structure: speech rows are their own labels, text rows are masked, the end marker is learned on channel 0
        a0 = total - audio - len(StubTokenizer().encode("<|end_of_speech|>"))
        assert (labels[:a0] == -100).all() and np.array_equal(labels[a0:a0 + audio], ids[a0:a0 + audio])
        assert np.array_equal(labels[a0 + audio:, 0], ids[a0 + audio:, 0]) and (labels[a0 + audio:, 1:] == -100).all()
        assert (ids[a0:a0 + audio, 0] >= 151665).all()
    with pytest.raises(ValueError, match=fx["no_audio_error"]):
        dp.process_inputs(StubTokenizer(), StubSpt({}), "p", "t", "cpu")


def test_process_data_writes_reference_file_formats(tmp_path, dp):
    """Both JSONL formats, skipped items, and the files the fine-tune dataset reads: `<name>.pkl` = one pickle per entry
    back to back, `<name>_metas.npy` = stack([byte offsets, total lengths, audio lengths])
    (reference finetune/data_preprocess.py:272-291)."""
    import generation_utils as gu
    rng = np.random.default_rng(0)
    lens = {"a.wav": 16000, "ref.wav": 6400, "main.wav": 12800}
    table = {}
    for name, n in lens.items():
        gu.save_wav(str(tmp_path / name), torch.from_numpy(rng.uniform(-0.5, 0.5, (1, n)).astype(np.float32)), 16000)
        table[n] = rng.integers(0, 1024, (8, n // 1280)).astype(np.int64)
    items = [
        {"file_path": str(tmp_path / "a.wav"), "full_transcript": "[S1]你好！[S2]Hello?"},
        {"file_path": str(tmp_path / "missing.wav"), "full_transcript": "x"},                      # skipped
        {"reference_audio": str(tmp_path / "ref.wav"), "reference_text": "[S1]ref.", "audio": str(tmp_path / "main.wav"), "text": "[S2]main."},
        {"text": "neither format"},                                                                 # skipped
        {"file_path": "", "full_transcript": "empty path"},                                         # skipped
    ]
    jl = tmp_path / "d.jsonl"
    jl.write_text("\n".join(json.dumps(it, ensure_ascii=False) for it in items) + "\n", encoding="utf-8")
    spt = StubSpt(table)
    dp.process_data(str(jl), "unused", str(tmp_path / "out"), data_name="train", use_normalize=True,
                    tokenizer=StubTokenizer(), spt=spt, device="cpu")
    metas = np.load(tmp_path / "out" / "train_metas.npy")
    assert metas.shape == (3, 2)
    entries = []
    with open(tmp_path / "out" / "train.pkl", "rb") as f:       # a file this test wrote itself
        for off in metas[0]:
            f.seek(int(off))
            entries.append(pickle.load(f))
    assert metas[0, 0] == 0 and metas[0, 1] > 0
    for e, total, audio in zip(entries, metas[1], metas[2]):
        ids, labels = np.array(e["input_ids"]), np.array(e["labels"])
        assert ids.shape == labels.shape == (total, 8)
        assert int((labels[:, 1] != -100).sum()) == audio
    assert list(metas[2]) == [12, 5 + 10]
    # normalisation + speaker tags reached the tokenizer: "[S1]你好！" -> "<speaker1>你好。"
    text0 = "".join(chr(t - 3) for t in np.array(entries[0]["input_ids"])[:, 0] if t < 151000)
    assert "<speaker1>你好，<speaker2>Hello." in text0 or "<speaker1>你好。" in text0
    assert spt.calls == [[16000], [6400], [12800]]
