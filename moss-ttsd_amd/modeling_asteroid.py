"""Drop-in for the reference's `modeling_asteroid` module on MI355X.

Keeps the call surface `generation_utils` / `inference.py` use
(reference modeling_asteroid.py: AsteroidTTSConfig :17-28, AsteroidTTSInstruct :288,
`.from_pretrained`, `.eval()`, `.to(device)`, `.generate(input_ids, attention_mask)`,
`.config`), and routes everything that computes to libmtts.so (hand-written HIP for
gfx950) through mtts.engine.Engine.  There is no PyTorch forward here and no CPU path.
"""
from __future__ import annotations

import glob
import json
import os

import numpy as np
import torch

from mtts import synth
from mtts.engine import Engine

_CFG_KEYS = ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
             "num_key_value_heads", "head_dim", "rms_norm_eps", "rope_theta", "max_position_embeddings",
             "channels", "speech_pad_token", "speech_vocab_size", "speech_token_range", "eos_token_id", "pad_token_id")


class AsteroidTTSConfig:
    """Qwen3Config fields + channels / speech_pad_token / speech_vocab_size / speech_token_range."""
    model_type = "asteroid_tts"

    def __init__(self, channels=8, speech_pad_token=1024, speech_vocab_size=1025, speech_token_range=(), **kw):
        base = synth.make_config()
        base.update(channels=channels, speech_pad_token=speech_pad_token, speech_vocab_size=speech_vocab_size)
        if speech_token_range:
            base["speech_token_range"] = list(speech_token_range)
        if "rope_parameters" in kw and isinstance(kw["rope_parameters"], dict):
            kw.setdefault("rope_theta", kw["rope_parameters"].get("rope_theta", base["rope_theta"]))
        if kw.get("head_dim") is None and "hidden_size" in kw and "num_attention_heads" in kw:
            kw["head_dim"] = kw["hidden_size"] // kw["num_attention_heads"]
        for k, v in kw.items():
            if k in base and v is not None:
                base[k] = v
        self.__dict__.update(base)
        self._extra = {k: v for k, v in kw.items() if k not in base}

    def to_dict(self):
        return {k: getattr(self, k) for k in _CFG_KEYS}

    @classmethod
    def from_pretrained(cls, path):
        with open(os.path.join(path, "config.json")) as f:
            return cls(**json.load(f))


class GenerationConfig:
    """The fields of generation_config.json the decode loop reads (modeling_asteroid.py:66-109)."""

    def __init__(self, **kw):
        self.max_new_tokens = kw.get("max_new_tokens")
        self.max_length = kw.get("max_length", 20)
        self.do_sample = bool(kw.get("do_sample", False))
        self.do_samples = kw.get("do_samples")
        self.layers = kw.get("layers")
        self.eos_token_id = kw.get("eos_token_id")
        self.temperature = kw.get("temperature")
        self.top_k = kw.get("top_k")
        self.top_p = kw.get("top_p")
        self.repetition_penalty = kw.get("repetition_penalty")
        self.seed = kw.get("seed")

    @classmethod
    def from_pretrained(cls, path):
        p = os.path.join(path, "generation_config.json")
        if not os.path.exists(p):
            return cls()
        with open(p) as f:
            return cls(**json.load(f))

    def channel_settings(self, channels):
        """-> (layers[8], do_samples[8]); the global-processor branch (:107-109) repeats one config."""
        if self.do_samples is not None:
            layers = list(self.layers or [])
            layers += [{}] * (channels - len(layers))
            return layers, list(self.do_samples)
        one = {}
        if self.do_sample:
            one = dict(repetition_penalty=self.repetition_penalty, temperature=self.temperature,
                       top_k=self.top_k, top_p=self.top_p)
        elif self.repetition_penalty not in (None, 1.0):
            one = dict(repetition_penalty=self.repetition_penalty)
        return [one] * channels, [self.do_sample] * channels


def _load_safetensors_dir(path):
    from safetensors.torch import load_file
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {path}")
    sd = {}
    for f in files:
        sd.update(load_file(f))
    return sd


class AsteroidTTSInstruct:
    MAX_ENGINE_BATCH = 128          # rows one engine pass carries (4 activation tiles share each weight stream)

    def __init__(self, config: AsteroidTTSConfig, state_dict=None, generation_config=None):
        self.config = config
        self.generation_config = generation_config or GenerationConfig(eos_token_id=config.eos_token_id)
        self.channels = config.channels
        self._sd = state_dict
        self._engine = None
        self._engine_key = None
        self.device = torch.device("cpu")
        self.training = False
        self.dtype = "bf16"             # "fp32": the strict-parity engine (from_pretrained(torch_dtype=torch.float32)); "fp16": its kernels with fp16 rounding points
        self.sample_seed = None         # explicit Philox key for the next generate() (tests); None = from torch's seed
        self.sample_rows = None         # Philox row id of each row of the next generate() (a rank's share of a sharded
                                        # batch sets its rows' job-wide positions); None = 0..B-1
        self._calls = 0

    # ---- loading -----------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, model_path, torch_dtype=torch.bfloat16, attn_implementation=None, **_):
        if torch_dtype not in (torch.bfloat16, torch.float32, torch.float16, None):
            raise NotImplementedError("the MI355X engine is built for the three dtypes inference.py offers (bf16, fp16, fp32); "
                                      f"torch_dtype={torch_dtype} is not one of them")
        if not os.path.isdir(model_path):
            raise FileNotFoundError(f"{model_path}: local checkpoint directory required (no network here)")
        cfg = AsteroidTTSConfig.from_pretrained(model_path)
        m = cls(cfg, _load_safetensors_dir(model_path), GenerationConfig.from_pretrained(model_path))
        m.dtype = {torch.float32: "fp32", torch.float16: "fp16"}.get(torch_dtype, "bf16")
        return m

    @classmethod
    def from_state_dict(cls, cfg_dict, state_dict, generation_config=None, dtype="bf16"):
        m = cls(AsteroidTTSConfig(**cfg_dict), state_dict, generation_config)
        m.dtype = dtype
        return m

    def eval(self):
        self.training = False
        return self

    def to(self, device):
        self.device = torch.device(device)
        return self

    def is_speech_token(self, tokens):
        lo, hi = self.config.speech_token_range
        return (tokens >= lo) & (tokens < hi)

    # ---- engine ------------------------------------------------------------------
    def _get_engine(self, batch, need_len):
        if self.device.type != "cuda":
            raise RuntimeError("AsteroidTTSInstruct on MI355X needs model.to('cuda'): the HIP engine has no CPU path")
        cap_len = max(4096, int(need_len)) + 64
        slots = 32 if batch <= 32 else (64 if batch <= 64 else self.MAX_ENGINE_BATCH)
        if self._engine is not None and self._engine_key[:2] == (str(self.device), cap_len) and self._engine_key[2] >= slots:
            return self._engine
        key = (str(self.device), cap_len, slots)
        if self._engine is None or self._engine_key != key:
            if self._engine is not None:
                self._engine.close()
            self._engine = Engine(self.config.to_dict(), max_batch=slots, max_seq_len=cap_len,
                                  device=str(self.device), dtype=self.dtype)
            self._engine.bind_state_dict(self._sd)
            self._engine_key = key
        return self._engine

    @torch.no_grad()
    def generate(self, input_ids=None, attention_mask=None, max_new_tokens=None, max_length=None, seed=None, **_):
        """LongTensor[B,T,8], mask[B,T] -> LongTensor[B, T-7+G, 8] (generation_utils.py:406-409)."""
        gc = self.generation_config
        B, T, C = input_ids.shape
        if C != self.channels:
            raise ValueError(f"Expected {self.channels} channels, got {C}")
        if attention_mask is None:
            attention_mask = torch.ones(B, T)
        # HF: max_length = max_new_tokens + input length when max_new_tokens is given
        mnt = max_new_tokens if max_new_tokens is not None else gc.max_new_tokens
        if max_length is None:
            max_length = (T + mnt) if mnt is not None else gc.max_length
        layers, do_samples = gc.channel_settings(C)
        seed = self._next_seed(seed)
        ids = input_ids.detach().cpu().numpy()
        msk = attention_mask.detach().cpu().numpy()
        rows = list(range(B)) if self.sample_rows is None else [int(r) for r in self.sample_rows]
        if len(rows) != B:
            raise ValueError(f"sample_rows has {len(rows)} entries for a batch of {B}")
        # room for the reference's finished-row flushes past max_length: 14 steps at least, the whole chain
        # (6 * B + 8, include/mtts.h: mtts_generate) where that is cheap
        eng = self._get_engine(B, int(max_length) + 6 * min(B, self.MAX_ENGINE_BATCH) + 8)
        if B <= self.MAX_ENGINE_BATCH:
            # one static batch, the reference's semantics: finished rows emit (eos, 1024 x 7) until the batch ends
            out = eng.generate(ids, msk, int(max_length), layers=layers, do_samples=do_samples, seed=seed, row_ids=rows)
        else:
            out = self._generate_scheduled(eng, ids, msk, int(max_length), layers, do_samples, seed, rows)
        return torch.from_numpy(out).to(input_ids.device)

    def _next_seed(self, seed):
        """Philox key of this call.  Explicit `seed=` / `generation_config.seed` / `model.sample_seed` win; otherwise it
        follows torch's global seed, which is what the reference's `inference.py --seed` sets through
        accelerate.set_seed (inference.py:69-72), advanced per call so that successive batches differ."""
        gc = self.generation_config
        if seed is None:
            seed = gc.seed if gc.seed is not None else self.sample_seed
        if seed is None:
            seed = (int(torch.initial_seed()) + 0x9E3779B97F4A7C15 * self._calls) & 0xFFFFFFFFFFFFFFFF
        self._calls += 1
        return int(seed)

    def _generate_scheduled(self, eng, ids, msk, max_length, layers, do_samples, seed, rows):
        """More rows than one pass carries: the continuous batcher serves them through MAX_ENGINE_BATCH slots (a
        finished dialogue's slot and KV pages go to the next one).  Row i draws from the Philox stream
        (seed; step, rows[i], channel) -- the stream row i of one static batch would use, so the same seed and prompts
        give the same tokens on either side of the 128-row limit (what differs: a dialogue cut off by max_length leaves
        at once here, while the reference keeps evaluating it and may resurrect it for a flush while another row is
        still flushing, modeling_asteroid.py:140-141,168)."""
        from mtts.scheduler import ContinuousBatcher
        B, T, C = ids.shape
        base = T - 7
        pads = [int(np.argmax(msk[b, :base] > 0)) for b in range(B)]
        prompts = [ids[b, pads[b]:] for b in range(B)]
        new = max_length - T
        cb = ContinuousBatcher(eng, slots=self.MAX_ENGINE_BATCH, gen_cap=max_length - base + 8, layers=layers,   # max_new + 7 flush steps
                               do_samples=do_samples)
        res = cb.run(prompts, new, seeds=[seed] * B, row_ids=rows)
        G = max(r.shape[0] - (T - pads[b] - 7) for b, r in enumerate(res))
        full = np.full((B, base + G, C), self.config.speech_pad_token, dtype=np.int64)
        full[:, :, 0] = self.config.eos_token_id            # finished-row padding (modeling_asteroid.py:155-158)
        for b, r in enumerate(res):
            full[b, :base] = ids[b, :base]
            gen = r[T - pads[b] - 7:]
            full[b, base:base + gen.shape[0]] = gen
        return full
