"""Drop-in for the reference's XY_Tokenizer (XY_Tokenizer/xy_tokenizer/model.py) on MI355X:
`load_from_checkpoint`, `.eval()`, `.to(device)`, `.decode(codes_list)`, sample-rate attributes.
Both branches run on the HIP codec engine (csrc/codec.hip): `.decode` (codes -> waveform) and
`.encode` (voice-clone prompt audio -> codes, log-mel front-end on the device)."""
from __future__ import annotations

import torch
import yaml

from mtts import synth_codec
from mtts.codec import CodecEngine


class XY_Tokenizer:
    def __init__(self, generator_params, state_dict=None):
        self.cfg = synth_codec.from_yaml_generator_params(generator_params)
        self.input_sample_rate = generator_params["input_sample_rate"]
        self.output_sample_rate = generator_params["output_sample_rate"]
        self.encoder_downsample_rate = 1280
        self.decoder_upsample_rate = 1920
        self.nq = generator_params["quantizer_kwargs"]["num_quantizers"]
        self._sd = state_dict
        self._engine = None
        self.device = torch.device("cpu")

    @classmethod
    def from_engine_config(cls, cfg, state_dict, input_sample_rate, output_sample_rate, nq):
        """A tokenizer from the engine-side config dict and a state dict that is already in memory (the ranks of a
        sharded job that did not read the checkpoint: inference_sharded.load_model_sharded)."""
        m = cls.__new__(cls)
        m.cfg = dict(cfg)
        m.input_sample_rate, m.output_sample_rate, m.nq = input_sample_rate, output_sample_rate, nq
        m.encoder_downsample_rate, m.decoder_upsample_rate = 1280, 1920
        m._sd, m._engine, m.device = state_dict, None, torch.device("cpu")
        return m

    @classmethod
    def load_from_checkpoint(cls, config_path: str, ckpt_path: str):
        with open(config_path) as f:
            config = yaml.safe_load(f)
        ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=True)
        sd = ckpt["generator"] if "generator" in ckpt else ckpt
        return cls(config["generator_params"], sd)

    def eval(self):
        return self

    def to(self, device):
        self.device = torch.device(device)
        return self

    def _get_engine(self):
        if self.device.type != "cuda":
            raise RuntimeError("XY_Tokenizer on MI355X needs spt.to('cuda'): the HIP codec has no CPU path")
        if self._engine is None:
            self._engine = CodecEngine(self.cfg, device=str(self.device))
            self._engine.bind_state_dict(self._sd)
        return self._engine

    @torch.inference_mode()
    def decode(self, codes_list, overlap_seconds=10, device=None):
        """B x LongTensor(nq,T) -> {"syn_wav_list": B x FloatTensor(1920*T,)}  (reference model.py:195-256)."""
        return {"syn_wav_list": self._get_engine().decode(codes_list, overlap_seconds=overlap_seconds)}

    @torch.inference_mode()
    def decode_each(self, codes_list, overlap_seconds=10, device=None):
        """Like one `decode([codes])` call per sequence (what the reference's process_batch does), executed together:
        windows of equal length from all sequences share codec calls (mtts/codec.py: CodecEngine.decode_each)."""
        return {"syn_wav_list": self._get_engine().decode_each(codes_list, overlap_seconds=overlap_seconds)}

    @torch.inference_mode()
    def encode(self, wav_list, overlap_seconds=10, device=None):
        """B x FloatTensor(T,) at 16 kHz -> {"codes_list": B x LongTensor(nq, T//1280)}  (reference model.py:131-192)."""
        return {"codes_list": self._get_engine().encode(wav_list, overlap_seconds=overlap_seconds)}
