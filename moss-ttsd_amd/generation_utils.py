"""Drop-in for the reference's `generation_utils` module (load_model / process_batch and the
delay-pattern helpers) with the model and the codec replaced by the MI355X engines.

Behaviour follows /root/reference/generation_utils.py: item parsing :27-87, prompt layout
:180-208, delay shift :211-218, left padding :221-237, un-shift :416-425, valid-length search
:240-249 (channel 1), text normalisation :252-338, the None-on-failure convention :434-467 (the codec
decodes the windows of all samples together; the reference calls it once per sample).  torchaudio is not required: wav I/O uses the stdlib.
"""
from __future__ import annotations

import os
import re
import traceback
import wave

import numpy as np
import torch

from modeling_asteroid import AsteroidTTSInstruct
from XY_Tokenizer.xy_tokenizer.model import XY_Tokenizer

MAX_CHANNELS = 8
SILENCE_DURATION = 0.0
SPEECH_OFFSET = 151665
SPEECH_PAD = 1024


def load_model(model_path, spt_config_path, spt_checkpoint_path, torch_dtype=torch.bfloat16,
               attn_implementation="flash_attention_2"):
    from transformers import AutoTokenizer
    tokenizer = AutoTokenizer.from_pretrained(model_path)
    model = AsteroidTTSInstruct.from_pretrained(model_path, torch_dtype=torch_dtype,
                                                attn_implementation=attn_implementation)
    spt = XY_Tokenizer.load_from_checkpoint(config_path=spt_config_path, ckpt_path=spt_checkpoint_path)
    return tokenizer, model.eval(), spt.eval()


# ---- item parsing -------------------------------------------------------------------
def _join(base, p):
    return os.path.join(base, p) if (isinstance(p, str) and base and p) else p


def _has_audio(v):
    return bool(v) if isinstance(v, str) else isinstance(v, tuple)


def process_jsonl_item(item):
    base = item.get("base_path", "")
    out = {"text": item.get("text", ""), "prompt_text": "", "prompt_audio": None}
    if "prompt_audio" in item and "prompt_text" in item:
        print("Using prompt_audio and prompt_text directly from item.")
        if item["prompt_audio"]:
            out["prompt_audio"] = _join(base, item["prompt_audio"])
            out["prompt_text"] = item["prompt_text"]
        return out
    a1, a2 = item.get("prompt_audio_speaker1", ""), item.get("prompt_audio_speaker2", "")
    if _has_audio(a1) or _has_audio(a2):
        print("Using speaker1 and speaker2 information for prompt audio and text.")
        out["prompt_audio"] = {"speaker1": _join(base, a1), "speaker2": _join(base, a2)}
    text = ""
    for tag, key in (("[S1]", "prompt_text_speaker1"), ("[S2]", "prompt_text_speaker2")):
        if item.get(key, ""):
            text += tag + item[key]
    out["prompt_text"] = text.strip()
    return out


# ---- audio loading (prompt audio; stdlib wav reader, linear-phase windowed-sinc resampler) --------
def _read_wav(path):
    """RIFF/WAVE reader standing in for torchaudio.load (reference generation_utils.py:96): -> (float32 [channels, n]
    in [-1, 1), sample_rate).  PCM 8/16/24/32-bit and IEEE float32, plain or WAVE_FORMAT_EXTENSIBLE.  The `data`
    chunk is read by its own size (clipped to the file), not by the RIFF header's: some of the reference's example
    files carry a RIFF size a few bytes short of the file, which the stdlib `wave` module would truncate by."""
    import struct
    with open(path, "rb") as f:
        d = f.read()
    if len(d) < 12 or d[:4] != b"RIFF" or d[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    fmt = data = None
    i = 12
    while i + 8 <= len(d):
        cid, sz = d[i:i + 4], struct.unpack("<I", d[i + 4:i + 8])[0]
        body = d[i + 8:i + 8 + sz]
        if cid == b"fmt ":
            fmt = body
        elif cid == b"data":
            data = body
            break
        i += 8 + sz + (sz & 1)
    if fmt is None or data is None or len(fmt) < 16:
        raise ValueError(f"{path}: missing fmt/data chunk")
    tag, ch, sr, _, _, bits = struct.unpack("<HHIIHH", fmt[:16])
    if tag == 0xFFFE and len(fmt) >= 26:
        tag = struct.unpack("<H", fmt[24:26])[0]
    frame = ch * (bits // 8)
    if ch < 1 or frame == 0:
        raise ValueError(f"{path}: bad fmt chunk")
    data = data[:len(data) // frame * frame]
    if tag == 1 and bits == 16:
        x = np.frombuffer(data, dtype="<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 8:
        x = (np.frombuffer(data, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif tag == 1 and bits == 24:
        b3 = np.frombuffer(data, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b3[:, 0] | (b3[:, 1] << 8) | (b3[:, 2] << 16)
        x = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
    elif tag == 1 and bits == 32:
        x = (np.frombuffer(data, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    elif tag == 3 and bits == 32:
        x = np.frombuffer(data, dtype="<f4").astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported wav encoding (format tag {tag}, {bits} bits)")
    return torch.from_numpy(x.reshape(-1, ch).T.copy()), int(sr)


def _resample(wav, sr, target, width=6, rolloff=0.99):
    """Windowed-sinc polyphase resampling with torchaudio.functional.resample's published defaults
    (sinc_interp_hann, lowpass_filter_width=6, rolloff=0.99).  torchaudio is absent here, so this
    path is PARITY UNPINNED (tests/test_io_cpu.py checks its length rule and gain); it feeds the prompt encoder."""
    import math
    g = math.gcd(int(sr), int(target))
    o, n = int(sr) // g, int(target) // g
    base = min(o, n) * rolloff
    w = math.ceil(width * o / base)
    idx = torch.arange(-w, w + o, dtype=torch.float64)[None, None] / o
    t = (torch.arange(0, -n, -1, dtype=torch.float64)[:, None, None] / n + idx) * base
    t = t.clamp(-width, width)
    win = torch.cos(t * math.pi / width / 2) ** 2
    t = t * math.pi
    k = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * win * (base / o)
    k = k.float()
    x = torch.nn.functional.pad(wav[:, None], (w, w + o))
    y = torch.nn.functional.conv1d(x, k, stride=o).transpose(1, 2).reshape(wav.shape[0], -1)
    return y[..., :math.ceil(n * wav.shape[-1] / o)]


def _load_single_audio(audio_input):
    if isinstance(audio_input, tuple) and len(audio_input) == 2:
        return audio_input
    if isinstance(audio_input, str):
        return _read_wav(audio_input)
    raise ValueError(f"Unsupported audio input format: {type(audio_input)}")


def _mono_16k(wav, sr, target):
    if sr != target:
        wav = _resample(wav, sr, target)
    if wav.shape[0] > 1:
        wav = wav.mean(dim=0, keepdim=True)
    return wav if wav.dim() == 2 else wav.unsqueeze(0)


def merge_speaker_audios(wav1, sr1, wav2, sr2, target_sample_rate=16000):
    return torch.cat([_mono_16k(wav1, sr1, target_sample_rate), _mono_16k(wav2, sr2, target_sample_rate)], dim=1)


def load_audio_data(prompt_audio, target_sample_rate=16000):
    if prompt_audio is None:
        return None
    try:
        if isinstance(prompt_audio, dict) and "speaker1" in prompt_audio and "speaker2" in prompt_audio:
            w1, s1 = _load_single_audio(prompt_audio["speaker1"])
            w2, s2 = _load_single_audio(prompt_audio["speaker2"])
            return merge_speaker_audios(w1, s1, w2, s2, target_sample_rate)
        return _mono_16k(*_load_single_audio(prompt_audio), target_sample_rate)
    except Exception as e:
        print(f"Error loading audio data: {e}")
        raise


# ---- prompt layout -------------------------------------------------------------------
def process_inputs(tokenizer, spt, prompt, text, device, audio_data=None, max_channels=8, pad_token=1024):
    seq = f"<|begin_of_style|>{prompt}<|end_of_style|>\n<|begin_of_text|>{text}<|end_of_text|>\n<|begin_of_speech|>"
    text_ids = np.array(tokenizer.encode(seq))
    ids = np.full((text_ids.shape[0], max_channels), pad_token)
    ids[:, 0] = text_ids
    if audio_data is None:
        return ids
    try:
        wav = torch.cat([audio_data, torch.zeros(audio_data.shape[0], int(SILENCE_DURATION * 16000))], dim=1)
        with torch.no_grad():
            codes = spt.encode([wav.squeeze().to(device)])["codes_list"][0].permute(1, 0).cpu().numpy()
        codes[:, 0] = codes[:, 0] + SPEECH_OFFSET
        return np.concatenate([ids, codes])
    except Exception as e:
        print(f"Error processing audio data: {e}")
        raise


def shifting_inputs(input_ids, tokenizer, pad_token=1024, max_channels=8):
    n = input_ids.shape[0]
    out = np.full((n + max_channels - 1, max_channels), pad_token, dtype=np.int64)
    out[:, 0] = tokenizer.pad_token_id
    for c in range(max_channels):
        out[c:n + c, c] = input_ids[:, c]
    return out


def rpadding(input_ids, channels, tokenizer):
    longest = max(x.shape[0] for x in input_ids)
    ids = np.full((len(input_ids), longest, channels), SPEECH_PAD)
    ids[:, :, 0] = tokenizer.pad_token_id
    mask = np.zeros((len(input_ids), longest))
    for b, x in enumerate(input_ids):
        ids[b, longest - x.shape[0]:] = x
        mask[b, longest - x.shape[0]:] = 1.0
    return torch.tensor(ids), torch.tensor(mask)


def find_max_valid_positions(C: torch.Tensor, invalid_value=1024) -> torch.Tensor:
    valid = C[:, :, 1] != invalid_value
    last = C.size(1) - 1 - torch.argmax(valid.flip(dims=[1]).int(), dim=1)
    return torch.where(valid.any(dim=1), last, -1)


# ---- text normalisation ----------------------------------------------------------------
_DECOR = "【】《》（）『』「」""\"-“”～~"
_PUNCT = str.maketrans({'！': '，', '!': ',', '；': '，', ';': ',', '：': '，', ':': ',', '、': '，', '？': '，', '?': ','})


def _normalize_segment(content):
    content = re.sub(f"[{re.escape(_DECOR)}]", "", content)
    content = re.sub(r'哈{2,}', '(笑)', content)
    content = re.sub(r'\b(ha(\s*ha)+)\b', '(laughs)', content, flags=re.IGNORECASE)
    content = content.replace('——', '，').replace('……', '，').translate(_PUNCT).strip()
    if len(content) > 1:
        tail = {"，": "。", ",": "."}.get(content[-1], content[-1])
        content = content[:-1].replace('。', '，') + tail
    return content


def normalize_text(text: str) -> str:
    text = re.sub(r'\[(\d+)\]', r'[S\1]', text)
    text = re.sub(r'\[(?!S\d+\])([^\]]*)\]', r'\1', text)
    parts = []
    for seg in re.split(r'(?=\[S\d+\])', text.replace("\n", " ")):
        seg = seg.strip()
        if not seg:
            continue
        m = re.match(r'^(\[S\d+\])\s*(.*)', seg)
        tag, content = m.groups() if m else ('', seg)
        parts.append([tag, _normalize_segment(content)])
    if not parts:
        return ""
    merged = [parts[0]]
    for tag, content in parts[1:]:
        if tag == merged[-1][0] and tag:
            merged[-1][1] += content
        else:
            merged.append([tag, content])
    return "".join(f"{t}{c}".strip() for t, c in merged).replace('‘', "'").replace('’', "'")


# ---- the batch API -------------------------------------------------------------------------
def process_batch(batch_items, tokenizer, model, spt, device, system_prompt, start_idx, use_normalize=False, indices=None):
    """-> (actual_texts_data, audio_results); a failed sample yields None, a batch-level failure re-raises.
    `indices` (not in the reference; used by inference_sharded.process_batch_sharded): the job-wide index of each item
    when this call serves one rank's share of a larger batch; default start_idx + position, as in the reference."""
    try:
        n = len(batch_items)
        gidx = [start_idx + i for i in range(n)] if indices is None else [int(x) for x in indices]
        print(f"Processing {n} samples starting from index {start_idx}...")
        texts, audios, meta = [], [], []
        for i, item in enumerate(batch_items):
            it = process_jsonl_item(item)
            original = it["prompt_text"] + it["text"] if it["prompt_text"] else it["text"]
            full = normalize_text(original) if use_normalize else original
            final = full.replace("[S1]", "<speaker1>").replace("[S2]", "<speaker2>")
            texts.append(final)
            audios.append(it["prompt_audio"])
            meta.append({"index": gidx[i], "original_text": original,
                         "normalized_text": normalize_text(original) if use_normalize else None,
                         "final_text": final, "use_normalize": use_normalize})
        seqs = []
        for text, audio in zip(texts, audios):
            audio_data = load_audio_data(audio) if audio else None
            seqs.append(shifting_inputs(process_inputs(tokenizer, spt, system_prompt, text, device, audio_data), tokenizer))
        input_ids, attention_mask = rpadding(seqs, MAX_CHANNELS, tokenizer)
        print("Starting batch audio generation...")
        start = input_ids.shape[1] - MAX_CHANNELS + 1
        outputs = model.generate(input_ids=input_ids.to(device), attention_mask=attention_mask.to(device))
        print(f"Original outputs shape: {outputs.shape}")
        outputs = outputs[:, start:]
        seq_len = outputs.shape[1] - MAX_CHANNELS + 1
        speech_ids = torch.zeros((outputs.shape[0], seq_len, MAX_CHANNELS), dtype=outputs.dtype, device=outputs.device)
        for j in range(MAX_CHANNELS):
            speech_ids[..., j] = outputs[:, j:seq_len + j, j]
        speech_ids[..., 0] -= SPEECH_OFFSET
        last = find_max_valid_positions(speech_ids)
        results = [None] * n
        valid = []
        for i in range(n):
            if int(last[i]) + 1 <= 0:
                print(f"Sample {gidx[i]} has no valid speech tokens")
            else:
                valid.append(i)
                print(f"Speech token shape for sample {gidx[i]}: {speech_ids[i, :int(last[i]) + 1].shape}")

        def pack(i, wav):
            wav = wav.cpu().detach()
            return {"audio_data": wav.unsqueeze(0) if wav.ndim == 1 else wav,
                    "sample_rate": spt.output_sample_rate, "index": gidx[i]}

        # The reference decodes one sample per spt.decode call (generation_utils.py:434-450).  Here the 30 s windows
        # of ALL samples go through the codec together, each exactly as in its own call (`decode_each`: only windows
        # of equal length share a call, nothing is padded); if that batched call fails, fall back to one call per
        # sample so that a bad sample still only costs its own entry (None).
        wavs = None
        if valid and hasattr(spt, "decode_each"):
            try:
                wavs = spt.decode_each([speech_ids[i, :int(last[i]) + 1].permute(1, 0) for i in valid],
                                       overlap_seconds=10)["syn_wav_list"]
            except Exception as e:
                print(f"Batched codec decode failed ({str(e)}); decoding sample by sample...")
        for k, i in enumerate(valid):
            try:
                wav = wavs[k] if wavs is not None else \
                    spt.decode([speech_ids[i, :int(last[i]) + 1].permute(1, 0)], overlap_seconds=10)["syn_wav_list"][0]
                results[i] = pack(i, wav)
                print(f"Audio generation completed: sample {gidx[i]}")
            except Exception as e:
                print(f"Error processing sample {gidx[i]}: {str(e)}, skipping...")
                traceback.print_exc()
                results[i] = None
        return meta, results
    except Exception as e:
        print(f"Error during batch processing: {str(e)}")
        raise


def save_wav(path, audio, sample_rate):
    """PCM16 writer standing in for torchaudio.save (reference inference.py:107-111)."""
    x = (audio.squeeze(0).clamp(-1, 1) * 32767.0).round().to(torch.int16).numpy()
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(int(sample_rate))
        w.writeframes(x.tobytes())
