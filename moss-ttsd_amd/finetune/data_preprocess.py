"""Fine-tune data path on the MI355X codec: JSONL of (audio, transcript) -> (input_ids, labels) rows.

Drop-in for the reference's `finetune/data_preprocess.py`: `process_inputs` :26-147 (segment layout, the 151665 offset on
channel 0, channel alignment, -100 label masks, the learned <|end_of_speech|> label, single-audio and
reference+main formats), `process_data` :149-300 (both JSONL formats, the one-pickle-per-entry file and the
[pointers, token lengths, audio lengths] metadata array the fine-tune dataset reads).  What computes -- `spt.encode` --
is the HIP encoder (csrc/codec.hip: mtts_codec_tokenize) through the XY_Tokenizer mirror; nothing here runs a model on
the CPU.  Training itself (finetune.py, HF Trainer) is out of scope.
"""
from __future__ import annotations

import argparse
import json
import os
import pickle
import sys

import numpy as np
import torch

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from generation_utils import load_audio_data, normalize_text  # noqa: E402

MODEL_PATH = "fnlp/MOSS-TTSD-v0.5"
SYSTEM_PROMPT = ("You are a speech synthesizer that generates natural, realistic, and human-like conversational audio "
                 "from dialogue text.")
SPT_CONFIG_PATH = "XY_Tokenizer/config/xy_tokenizer_config.yaml"
SPT_CHECKPOINT_PATH = "XY_Tokenizer/weights/xy_tokenizer.ckpt"
MAX_CHANNELS = 8
SILENCE_DURATION = 0.0
SPEECH_OFFSET = 151665
IGNORE = -100


def load_tokenizer(model_path, spt_config_path, spt_checkpoint_path):
    from transformers import AutoTokenizer
    from XY_Tokenizer.xy_tokenizer.model import XY_Tokenizer
    return AutoTokenizer.from_pretrained(model_path), XY_Tokenizer.load_from_checkpoint(
        config_path=spt_config_path, ckpt_path=spt_checkpoint_path).eval()


def _text_rows(token_ids, channels, pad_token, learn=False):
    """Rows of a text segment: the ids on channel 0, `pad_token` elsewhere; labels -100, or the ids themselves on
    channel 0 when the segment is to be learned."""
    ids = np.full((len(token_ids), channels), pad_token)
    ids[:, 0] = token_ids
    labels = np.full(ids.shape, IGNORE)
    if learn:
        labels[:, 0] = ids[:, 0]
    return ids, labels


def _encode(spt, wav, device):
    """One waveform [1, n] (or [n]) -> codes [frames, nq], with the (zero-length) trailing silence the reference appends."""
    wav = wav[None] if wav.dim() == 1 else wav
    wav = torch.cat([wav, torch.zeros(wav.shape[0], int(SILENCE_DURATION * 16000))], dim=1)
    with torch.no_grad():
        codes = spt.encode([wav.squeeze().to(device)])["codes_list"][0]
    return codes.permute(1, 0).cpu().numpy()


def process_inputs(tokenizer, spt, prompt, text, device, audio_data=None, reference_audio=None, main_audio=None,
                   max_channels=8, pad_token=1024):
    """-> (input_ids [L, 8], labels [L, 8], total_length, audio_length); reference finetune/data_preprocess.py:26-147."""
    if reference_audio is not None and main_audio is not None:
        try:        # two recordings, encoded separately and joined at token level
            audio = np.concatenate([_encode(spt, reference_audio, device), _encode(spt, main_audio, device)], axis=0)
        except Exception as e:
            print(f"Error processing two audio files: {e}")
            raise
    elif audio_data is not None:
        try:
            audio = _encode(spt, audio_data, device)
        except Exception as e:
            print(f"Error processing audio data: {e}")
            raise
    else:
        audio = None
    head = [_text_rows(tokenizer.encode(f"<|begin_of_style|>{prompt}<|end_of_style|>\n<|begin_of_text|>"), max_channels, pad_token),
            _text_rows(tokenizer.encode(text, add_special_tokens=False), max_channels, pad_token),
            _text_rows(tokenizer.encode("<|end_of_text|>\n<|begin_of_speech|>"), max_channels, pad_token)]
    if audio is None:
        raise ValueError("No audio data provided")
    audio[:, 0] += SPEECH_OFFSET
    if audio.shape[1] != max_channels:                       # channel alignment: cut, or pad with the speech pad token
        fit = np.full((audio.shape[0], max_channels), pad_token)
        k = min(audio.shape[1], max_channels)
        fit[:, :k] = audio[:, :k]
        audio = fit
    tail = _text_rows(tokenizer.encode("<|end_of_speech|>"), max_channels, pad_token, learn=True)
    parts = head + [(audio, audio.copy()), tail]             # speech rows are their own labels
    input_ids = np.concatenate([p[0] for p in parts])
    labels = np.concatenate([p[1] for p in parts])
    return input_ids, labels, input_ids.shape[0], audio.shape[0]


def _item_to_rows(idx, item, tokenizer, spt, device, use_normalize):
    """One JSONL item -> process_inputs result, or None (with the reference's warning) when it has to be skipped."""
    def final(text):
        text = normalize_text(text) if use_normalize else text
        return text.replace("[S1]", "<speaker1>").replace("[S2]", "<speaker2>")

    if "file_path" in item and "full_transcript" in item:
        path = item["file_path"]
        if not path:
            print(f"Warning: Item {idx} has empty file_path, skipping...")
            return None
        if not os.path.exists(path):
            print(f"Warning: Audio file not found: {path}, skipping item {idx}...")
            return None
        try:
            audio = load_audio_data(path)
        except Exception as e:
            print(f"Warning: Failed to load audio from {path}: {e}, skipping item {idx}...")
            return None
        return process_inputs(tokenizer, spt, SYSTEM_PROMPT, final(item["full_transcript"]), device, audio,
                              max_channels=MAX_CHANNELS)
    if all(k in item for k in ("reference_audio", "reference_text", "audio", "text")):
        ref_path, path = item["reference_audio"], item["audio"]
        if not ref_path or not path:
            print(f"Warning: Item {idx} has empty audio paths, skipping...")
            return None
        if not os.path.exists(ref_path):
            print(f"Warning: Reference audio file not found: {ref_path}, skipping item {idx}...")
            return None
        if not os.path.exists(path):
            print(f"Warning: Audio file not found: {path}, skipping item {idx}...")
            return None
        try:
            return process_inputs(tokenizer, spt, SYSTEM_PROMPT, final(item["reference_text"] + item["text"]), device,
                                  reference_audio=load_audio_data(ref_path), main_audio=load_audio_data(path),
                                  max_channels=MAX_CHANNELS)
        except Exception as e:
            print(f"Warning: Failed to load audio files: {e}, skipping item {idx}...")
            return None
    print(f"Warning: Item {idx} missing required fields for both supported formats, skipping...")
    return None


def process_data(jsonl, model_path, output_dir, data_name="processd_data", use_normalize=True, tokenizer=None, spt=None,
                 device=None):
    """JSONL -> `<data_name>.pkl` (one pickle per entry, back to back) + `<data_name>_metas.npy` = stack([byte offsets,
    total lengths, audio lengths]); reference finetune/data_preprocess.py:149-300.  `tokenizer` / `spt` (not in the
    reference): already-loaded objects, so a caller need not load them twice."""
    os.makedirs(output_dir, exist_ok=True)
    device = device or ("cuda" if torch.cuda.is_available() else "cpu")
    print(f"Using device: {device}")
    if tokenizer is None or spt is None:
        print("Loading models...")
        tokenizer, spt = load_tokenizer(model_path, SPT_CONFIG_PATH, SPT_CHECKPOINT_PATH)
    spt = spt.to(device)
    try:
        with open(jsonl) as f:
            items = [json.loads(line) for line in f.readlines()]
        print(f"Loaded {len(items)} items from {jsonl}")
    except FileNotFoundError:
        print(f"Error: JSONL file '{jsonl}' not found")
        return
    except json.JSONDecodeError as e:
        print(f"Error parsing JSONL file: {e}")
        return
    entries, totals, audios = [], [], []
    for idx, item in enumerate(items):
        got = _item_to_rows(idx, item, tokenizer, spt, device, use_normalize)
        if got is None:
            continue
        ids, labels, total, n_audio = got
        entries.append({"input_ids": ids.tolist(), "labels": labels.tolist()})
        totals.append(total)
        audios.append(n_audio)
        print(f"Processed item {idx + 1}/{len(items)}: input_ids shape {ids.shape}, labels shape {labels.shape}, "
              f"total_len={total}, audio_len={n_audio}")
    offsets = []
    pkl_path = os.path.join(output_dir, f"{data_name}.pkl")
    with open(pkl_path, "wb") as f:
        for e in entries:
            offsets.append(f.tell())
            pickle.dump(e, f)
    meta_path = os.path.join(output_dir, f"{data_name}_metas.npy")
    np.save(meta_path, np.stack([np.array(offsets), np.array(totals), np.array(audios)]))
    print(f"Saved {len(entries)} processed items to {pkl_path}")
    print(f"Saved metadata (pointers, tokens_lengths, tims_lengths) to {meta_path}")
    print(f"Total sequences processed: {len(entries)}")
    if entries:
        print(f"Average total length: {np.mean(totals):.1f}, Average audio length: {np.mean(audios):.1f}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Fine-tune data preprocessing on the MI355X codec encoder")
    ap.add_argument("--jsonl", type=str, required=True)
    ap.add_argument("--model_path", type=str)
    ap.add_argument("--output_dir", type=str, required=True)
    ap.add_argument("--data_name", default="processed_data")
    ap.add_argument("--use_normalize", action="store_true", default=False)
    a = ap.parse_args()
    if not os.path.exists(a.jsonl):
        raise ValueError(f"JSONL file '{a.jsonl}' does not exist.")
    if not a.model_path:
        a.model_path = MODEL_PATH
    elif not os.path.exists(a.model_path):
        raise ValueError(f"Model path '{a.model_path}' does not exist.")
    process_data(a.jsonl, a.model_path, a.output_dir, a.data_name, a.use_normalize)
