"""Synthetic AsteroidTTS configs, weights and prompts (numpy only).

There is no checkpoint of fnlp/MOSS-TTSD-v0.5 in this environment, so tests,
fixtures and the bench all run on seeded random weights of the architecture
that `AsteroidTTSConfig` (reference modeling_asteroid.py:17-28) describes.
The generator is pure numpy so that the golden-fixture script (which imports
the reference), the numpy oracle and the HIP engine all see bit-identical
tensors from nothing but (config, seed).

Weight names follow the reference state dict
(`model.embedding_list.{c}.weight`, `model.language_model.layers.{n}...`,
reference modeling_asteroid.py:220-226,300-303; heads are tied to the
embeddings, :315-317, so only the embeddings are generated).
"""
from __future__ import annotations

import numpy as np

# Constants the reference hard-codes (modeling_asteroid.py:19-22,126,128;
# generation_utils.py:202,425).
CHANNELS = 8
SPEECH_PAD = 1024
SPEECH_VOCAB = 1025
SPEECH_OFFSET = 151665
EOS_ID = 152694
TEXT_VOCAB = 152697


def make_config(**over):
    """Dict with the fields of AsteroidTTSConfig/Qwen3Config the hot path reads."""
    cfg = dict(
        vocab_size=TEXT_VOCAB,
        hidden_size=2048,
        intermediate_size=6144,
        num_hidden_layers=28,
        num_attention_heads=16,
        num_key_value_heads=8,
        head_dim=128,
        rms_norm_eps=1e-6,
        rope_theta=1e6,
        max_position_embeddings=40960,
        channels=CHANNELS,
        speech_pad_token=SPEECH_PAD,
        speech_vocab_size=SPEECH_VOCAB,
        speech_token_range=[SPEECH_OFFSET, SPEECH_OFFSET + 1024],
        eos_token_id=EOS_ID,
        pad_token_id=151643,
    )
    cfg.update(over)
    return cfg


def assumed_1p7b():
    """ASSUMED Qwen3-1.7B-class dims (SURVEY.md reading notes / BASELINE.md §3)."""
    return make_config()


def tiny(**over):
    """Small config used by fixtures; keeps head_dim=128 and the real vocab ids
    (the reference indexes logits at 152694 and 1024 unconditionally)."""
    base = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                num_attention_heads=4, num_key_value_heads=2, head_dim=128,
                max_position_embeddings=1024)
    base.update(over)
    return make_config(**base)


def weight_shapes(cfg):
    """Ordered (name, shape, kind) list; order defines the RNG stream."""
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    D = cfg["head_dim"]
    nq, nkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    out = [("model.embedding_list.0.weight", (cfg["vocab_size"], H), "emb")]
    for c in range(1, cfg["channels"]):
        out.append((f"model.embedding_list.{c}.weight", (cfg["speech_vocab_size"], H), "emb"))
    for n in range(cfg["num_hidden_layers"]):
        p = f"model.language_model.layers.{n}."
        out += [
            (p + "input_layernorm.weight", (H,), "norm"),
            (p + "self_attn.q_proj.weight", (nq * D, H), "lin"),
            (p + "self_attn.k_proj.weight", (nkv * D, H), "lin"),
            (p + "self_attn.v_proj.weight", (nkv * D, H), "lin"),
            (p + "self_attn.q_norm.weight", (D,), "norm"),
            (p + "self_attn.k_norm.weight", (D,), "norm"),
            (p + "self_attn.o_proj.weight", (H, nq * D), "lin"),
            (p + "post_attention_layernorm.weight", (H,), "norm"),
            (p + "mlp.gate_proj.weight", (I, H), "lin"),
            (p + "mlp.up_proj.weight", (I, H), "lin"),
            (p + "mlp.down_proj.weight", (H, I), "lin"),
        ]
    out.append(("model.language_model.norm.weight", (H,), "norm"))
    return out


def round_bf16(x):
    """fp32 -> nearest-even bf16, returned as fp32 (numpy)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    out = ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)
    return np.where(np.isnan(x), x, out)


def synth_weights(cfg, seed, lin_std=None, emb_std=0.02, emb_row_sigma=0.0,
                  speech_boost=1.0, eos_boost=1.0, bf16=True):
    """name -> fp32 ndarray (values already bf16-representable when bf16=True).

    emb_row_sigma > 0 multiplies embedding/head rows by exp(sigma*N(0,1)): a
    heavy-tailed row norm gives the peaked logits a trained model has, which
    keeps greedy argmax margins far above one bf16 ulp in the fixtures.
    speech_boost / eos_boost scale the channel-0 rows of the speech range and of
    the EOS id: with plain N(0, s) rows channel 0 would almost never pick a
    speech token and every dialogue would flush after 7 steps.
    """
    rng = np.random.default_rng(seed)
    H = cfg["hidden_size"]
    if lin_std is None:
        lin_std = 1.0 / np.sqrt(H) if H <= 512 else 0.02
    w = {}
    for name, shape, kind in weight_shapes(cfg):
        if kind == "norm":
            a = 1.0 + 0.1 * rng.standard_normal(shape, dtype=np.float32)
        elif kind == "emb":
            a = emb_std * rng.standard_normal(shape, dtype=np.float32)
            if emb_row_sigma > 0:
                a *= np.exp(emb_row_sigma * rng.standard_normal((shape[0], 1), dtype=np.float32))
            if name.endswith("embedding_list.0.weight"):
                lo, hi = cfg["speech_token_range"]
                a[lo:hi] *= np.float32(speech_boost)
                a[cfg["eos_token_id"]] *= np.float32(eos_boost)
        else:
            a = np.float32(lin_std) * rng.standard_normal(shape, dtype=np.float32)
        a = a.astype(np.float32)
        w[name] = round_bf16(a) if bf16 else a
    return w


def shifting_inputs(input_ids, pad_token_id, pad_token=SPEECH_PAD, max_channels=CHANNELS):
    """Delay pattern (semantics of reference generation_utils.py:211-218)."""
    n = input_ids.shape[0]
    out = np.full((n + max_channels - 1, max_channels), pad_token, dtype=np.int64)
    out[:, 0] = pad_token_id
    for c in range(max_channels):
        out[c:n + c, c] = input_ids[:, c]
    return out


def left_pad(seqs, pad_token_id, channels=CHANNELS):
    """Left-pad a ragged batch (semantics of reference generation_utils.py:221-237)."""
    T = max(s.shape[0] for s in seqs)
    ids = np.full((len(seqs), T, channels), SPEECH_PAD, dtype=np.int64)
    ids[:, :, 0] = pad_token_id
    mask = np.zeros((len(seqs), T), dtype=np.float64)
    for b, s in enumerate(seqs):
        ids[b, T - s.shape[0]:] = s
        mask[b, T - s.shape[0]:] = 1.0
    return ids, mask


def synth_prompts(cfg, seed, batch, prompt_len, audio_frac=0.5, ragged=True):
    """SURVEY.md §8d prompts: text part on channel 0 (others 1024), optional
    audio part (ch0 = 151665+U[0,1024), ch1-7 U[0,1024)), delay-shifted and
    left-padded.  Returns (input_ids[B,T,8] int64, attention_mask[B,T] f64)."""
    rng = np.random.default_rng(seed)
    seqs = []
    for _ in range(batch):
        n = prompt_len - 7
        if ragged:
            n = int(rng.integers(max(2, n // 2), n + 1))
        n_audio = int(n * audio_frac)
        n_text = n - n_audio
        raw = np.full((n, CHANNELS), SPEECH_PAD, dtype=np.int64)
        raw[:n_text, 0] = rng.integers(0, 151643, n_text)
        if n_audio:
            raw[n_text:, 0] = SPEECH_OFFSET + rng.integers(0, 1024, n_audio)
            raw[n_text:, 1:] = rng.integers(0, 1024, (n_audio, CHANNELS - 1))
        seqs.append(shifting_inputs(raw, cfg["pad_token_id"]))
    return left_pad(seqs, cfg["pad_token_id"])
