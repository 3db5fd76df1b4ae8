"""normalize_text / process_jsonl_item fixtures from the REFERENCE's own functions
(/root/reference/generation_utils.py:27-87,:252-338).  Build container only."""
import json
import os
import sys
import types

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
import generation_utils as ref  # noqa: E402

texts = [
    "[1]你好！[2]哈哈哈，是吗？", "[S1]Hello: world; ok! [S1]again?", "no tags 【here】", "[note]x[S2]haha yes……no", "",
    "[S1]一。二。三。[S2]“引用”——破折号、顿号；分号：冒号", "[S1]a\n[S2]b\n\n[S1]c", "[S1] Ha ha ha! That's ‘great’.[S2]Hahaha",
    "[S3]三号说话人（括号）~波浪", "x", "[S1]", "[S1]。", "[S1]wow,", "[S2]哈", "[S1]A-B \"q\" 《书》", "[10]ten[2]two",
]
for fn in ("examples.jsonl", "examples_single_reference.jsonl", "examples_only_text.jsonl"):
    with open(os.path.join("/root/reference/examples", fn)) as f:
        for line in f:
            it = json.loads(line)
            for k in ("text", "prompt_text", "prompt_text_speaker1", "prompt_text_speaker2"):
                if it.get(k):
                    texts.append(it[k])
items = [
    {"text": "[S1]hi", "prompt_audio": "a.wav", "prompt_text": "[S1]p", "base_path": "/x"},
    {"text": "t", "prompt_audio_speaker1": "s1.wav", "prompt_text_speaker1": "one", "prompt_audio_speaker2": "s2.wav",
     "prompt_text_speaker2": "two", "base_path": "b"},
    {"text": "only"}, {"text": "e", "prompt_audio": "", "prompt_text": "zzz"},
    {"text": "q", "prompt_text_speaker1": "just text"}, {"text": "r", "prompt_audio_speaker2": "only2.wav"},
]
out = {"normalize": [[t, ref.normalize_text(t)] for t in texts],
       "items": [[it, ref.process_jsonl_item(dict(it))] for it in items]}
json.dump(out, open(os.path.join(HERE, "text_glue.json"), "w"), ensure_ascii=False, indent=0)
print(len(texts), "texts")
