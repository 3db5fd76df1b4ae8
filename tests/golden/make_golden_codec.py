"""Codec golden fixtures: run the REFERENCE XY_Tokenizer.decode on CPU with synthetic
weights (build container only; needs /root/reference).

    python tests/golden/make_golden_codec.py

Import-time shims only: empty `torchaudio` / `librosa` modules (nn/modules.py:8-21 imports
them for code the inference path never calls).  Only data is written: codes, (config, seed)
and a strided subset of the reference waveform plus per-second RMS.
"""
import json
import os
import sys
import types

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "moss-ttsd_amd"))
from mtts import synth_codec  # noqa: E402

import transformers  # noqa: F401,E402  (must be imported before the stubs)
for n in ["torchaudio", "torchaudio.functional", "torchaudio.functional.functional", "librosa"]:
    sys.modules[n] = types.ModuleType(n)
sys.modules["torchaudio.functional.functional"]._hz_to_mel = None
sys.modules["torchaudio.functional.functional"]._mel_to_hz = None
sys.modules["torchaudio"].functional = sys.modules["torchaudio.functional"]
sys.modules["torchaudio.functional"].functional = sys.modules["torchaudio.functional.functional"]
sys.path.insert(0, "/root/reference/XY_Tokenizer")
from xy_tokenizer.model import XY_Tokenizer  # noqa: E402

STRIDE = 13


def build_reference(cfg, w, encoder=False):
    gp = yaml.safe_load(open("/root/reference/XY_Tokenizer/config/xy_tokenizer_config.yaml"))["generator_params"]
    # when the encoder side is not run, shrink it so the module builds fast
    for k in ["semantic_encoder_kwargs", "acoustic_encoder_kwargs"]:
        gp[k]["encoder_layers"] = cfg["enc_layers"] if encoder else 1
    gp["semantic_encoder_adapter_kwargs"]["encoder_layers"] = cfg["sem_adapter_layers"] if encoder else 1
    gp["pre_rvq_adapter_kwargs"]["encoder_layers"] = cfg["pre_rvq_layers"] if encoder else 1
    gp["post_rvq_adapter_kwargs"]["encoder_layers"] = cfg["adapter_layers"]
    gp["acoustic_decoder_kwargs"]["decoder_layers"] = cfg["dec_layers"]
    gp["vocos_kwargs"]["num_layers"] = cfg["voc_layers"]
    m = XY_Tokenizer(gp).eval()
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    dec = ("quantizer.output_proj", "quantizer.quantizers", "post_rvq_adapter", "upsample", "acoustic_decoder", "enhanced_vocos")
    bad = [k for k in missing if k.startswith(dec) and not any(s in k for s in ("inited", "cluster_size", "embed_avg", "positional_embedding", "istft.window"))]
    assert not bad, bad
    if encoder:
        bad = [k for k in missing if not k.startswith(dec) and not any(s in k for s in ("positional_embedding",))]
        assert not bad, bad
        for q in m.quantizer.quantizers:       # codebooks are buffers that count as "not initialised" by default
            q.inited.fill_(True)
    return m


def make_encode(name, cfg, seed, lengths):
    """Reference XY_Tokenizer.encode (model.py:131-192) on synthetic audio -> code ids."""
    w = synth_codec.synth_weights(cfg, seed, encoder=True)
    m = build_reference(cfg, w, encoder=True)
    wavs = synth_codec.synth_wavs(seed + 1, lengths)
    with torch.no_grad():
        res = m.encode([torch.from_numpy(x) for x in wavs], overlap_seconds=10, device=torch.device("cpu"))
        # second-best distance margins of the first window (how safe each argmin is)
    d = dict(cfg=json.dumps(cfg), seed=seed, lengths=np.array(lengths))
    for i, cds in enumerate(res["codes_list"]):
        d[f"codes{i}"] = cds.numpy().astype(np.int16)
        print(name, i, tuple(cds.shape), cds[:, :6].tolist()[0])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)


def make(name, cfg, seed, lengths):
    w = synth_codec.synth_weights(cfg, seed)
    m = build_reference(cfg, w)
    codes = synth_codec.synth_codes(cfg, seed + 1, lengths)
    with torch.no_grad():
        res = m.decode([torch.from_numpy(c) for c in codes], overlap_seconds=10, device=torch.device("cpu"))
    d = dict(cfg=json.dumps(cfg), seed=seed, lengths=np.array(lengths), stride=STRIDE)
    for i, wv in enumerate(res["syn_wav_list"]):
        wv = wv.numpy().astype(np.float32)
        d[f"wav{i}_sub"] = wv[::STRIDE].copy()
        d[f"wav{i}_len"] = wv.shape[0]
        sec = cfg["output_sample_rate"]
        d[f"wav{i}_rms"] = np.array([np.sqrt(np.mean(wv[s:s + sec].astype(np.float64) ** 2)) for s in range(0, wv.shape[0], sec)])
        print(name, i, wv.shape, "rms", float(np.sqrt(np.mean(wv.astype(np.float64) ** 2))), "absmax", float(np.abs(wv).max()))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)


if __name__ == "__main__":
    torch.set_num_threads(8)
    red = synth_codec.reduced()
    if "enc" in sys.argv[1:]:
        make_encode("codec_enc_3s", red, 11, [48000])
        make_encode("codec_enc_ragged", red, 12, [80000, 33000])
        make_encode("codec_enc_35s", red, 13, [560000])
        sys.exit(0)
    make("codec_T40", red, 5, [40])
    make("codec_ragged_1win", red, 6, [375, 200])          # exactly one window + a padded row
    make("codec_T600", red, 7, [600])                      # 3 windows (starts 0, 250, 500)
    make("codec_full_T24", synth_codec.codec_config(), 8, [24])
    make_encode("codec_enc_3s", red, 11, [48000])
    make_encode("codec_enc_ragged", red, 12, [80000, 33000])          # batch with a shorter row
    make_encode("codec_enc_35s", red, 13, [560000])                   # 2 windows (30 s + stride 20 s)
