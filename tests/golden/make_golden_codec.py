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


def make_encode(name, cfg, seed, lengths, margins=False):
    """Reference XY_Tokenizer.encode (model.py:131-192) on synthetic audio -> code ids.
    margins=True also records, per code, how safe the reference's argmin was: forward hooks on the 8 VectorQuantize
    stages (observers only) recompute `dist` exactly as nn/quantizer.py:167-170 does from the stage's input and keep
    `second smallest - smallest` and the smallest itself; the per-window values are stitched like the codes are
    (model.py:170-184)."""
    w = synth_codec.synth_weights(cfg, seed, encoder=True)
    m = build_reference(cfg, w, encoder=True)
    wavs = synth_codec.synth_wavs(seed + 1, lengths)
    calls = []                                   # one entry per inference_tokenize call: list of nq x (gap, best) [B,T']
    hooks = []
    if margins:
        def hook(mod, args, out):
            z = args[0].float()
            enc = mod.in_project(z).float().permute(0, 2, 1).reshape(-1, mod.codebook.shape[1])
            cb = mod.codebook.float()
            dist = enc.pow(2).sum(1, keepdim=True) - 2 * enc @ cb.t() + cb.pow(2).sum(1, keepdim=True).t()
            two = torch.topk(dist, 2, dim=1, largest=False).values
            assert torch.equal((-dist).max(1)[1].reshape(z.shape[0], -1), out[3])       # the observer sees what the stage saw
            calls[-1].append(((two[:, 1] - two[:, 0]).reshape(z.shape[0], -1).numpy(), two[:, 0].reshape(z.shape[0], -1).numpy()))
        for q in m.quantizer.quantizers:
            hooks.append(q.register_forward_hook(hook))
        tok = m.inference_tokenize

        def tok_logged(x, ln):
            calls.append([])
            return tok(x, ln)
        m.inference_tokenize = tok_logged
    with torch.no_grad():
        res = m.encode([torch.from_numpy(x) for x in wavs], overlap_seconds=10, device=torch.device("cpu"))
    for h in hooks:
        h.remove()
    d = dict(cfg=json.dumps(cfg), seed=seed, lengths=np.array(lengths))
    for i, cds in enumerate(res["codes_list"]):
        d[f"codes{i}"] = cds.numpy().astype(np.int16)
        print(name, i, tuple(cds.shape), cds[:, :6].tolist()[0])
        if margins:
            keep = 20 * cfg["input_sample_rate"] // cfg["encoder_downsample_rate"]      # 250 codes kept per window
            gap = np.concatenate([np.stack([c[q][0][i, :keep] for q in range(cfg["nq"])]) for c in calls], axis=1)
            best = np.concatenate([np.stack([c[q][1][i, :keep] for q in range(cfg["nq"])]) for c in calls], axis=1)
            d[f"gap{i}"] = gap[:, :cds.shape[1]].astype(np.float32)
            d[f"best{i}"] = best[:, :cds.shape[1]].astype(np.float32)
            rel = d[f"gap{i}"] / np.abs(d[f"best{i}"])
            print(name, i, "argmin gap: min abs %.3g, min rel %.3g, share rel < 1e-4: %.4f"
                  % (d[f"gap{i}"].min(), rel.min(), (rel < 1e-4).mean()))
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
    full = synth_codec.codec_config()
    if "full" in sys.argv[1:]:
        # PRODUCTION DEPTH (round 3): 4 + 12 transformer layers, 30 ConvNeXt blocks, whole windows
        make("codec_full_T375", full, 21, [375])                          # exactly one full 30 s window (1 500 keys)
        make("codec_full_T520", full, 22, [520])                          # 3 windows (starts 0, 250, 500), ragged tail
        sys.exit(0)
    if "encfull" in sys.argv[1:]:
        # the two 12-layer encoders + 4-layer adapters: exact ids and the margin of every argmin
        make_encode("codec_enc_full_12s", full, 31, [192000], margins=True)
        make_encode("codec_enc_full_ragged", full, 32, [100000, 41000], margins=True)
        sys.exit(0)
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
