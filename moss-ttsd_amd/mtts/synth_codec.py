"""XY_Tokenizer decode-side configuration and synthetic weights (numpy only).

The real checkpoint (fnlp/XY_Tokenizer_TTSD_V0) is not in this environment; shapes
follow /root/reference/XY_Tokenizer/config/xy_tokenizer_config.yaml and the module
constructors (XY_Tokenizer/xy_tokenizer/model.py:39-49, nn/modules.py, nn/quantizer.py).
Names are the reference's state-dict names; weight-norm tensors (`weight_g`,
`weight_v`, nn/quantizer.py:10-11) are generated as such and folded by the loader.
"""
from __future__ import annotations

import numpy as np


def codec_config(**over):
    """Decode-side fields of xy_tokenizer_config.yaml (generator_params)."""
    cfg = dict(
        input_sample_rate=16000, output_sample_rate=24000,
        encoder_downsample_rate=1280, decoder_upsample_rate=1920,
        nq=8, codebook_size=1024, rvq_dim=512, quant_out_dim=3072,
        adapter_layers=4, adapter_dim=768, adapter_heads=12, adapter_ffn=3072, adapter_max_pos=375,
        up_stride=4,
        dec_layers=12, dec_dim=768, dec_heads=12, dec_ffn=3072, dec_max_pos=1500, mel_bins=80,
        voc_dim=512, voc_inter=4096, voc_layers=30, n_fft=960, hop=240,
        # encode side: mel front-end, two OmniAudioEncoders, adapters, down-conv, RVQ search
        mel_n_fft=400, mel_hop=160, mel_frames=3000, enc_layers=12, enc_dim=768, enc_heads=12, enc_ffn=3072,
        enc_max_pos=1500, sem_adapter_layers=4, pre_rvq_layers=4, down_pool=4,
    )
    cfg.update(over)
    return cfg


def reduced(**over):
    """Same widths, fewer layers: what the fixtures run (the kernels see real shapes)."""
    base = dict(adapter_layers=1, dec_layers=2, voc_layers=3, enc_layers=2, sem_adapter_layers=1, pre_rvq_layers=1)
    base.update(over)
    return codec_config(**base)


def from_yaml_generator_params(gp):
    q, a, d, v = gp["quantizer_kwargs"], gp["post_rvq_adapter_kwargs"], gp["acoustic_decoder_kwargs"], gp["vocos_kwargs"]
    return codec_config(
        input_sample_rate=gp["input_sample_rate"], output_sample_rate=gp["output_sample_rate"],
        nq=q["num_quantizers"], codebook_size=q["codebook_size"], rvq_dim=q["rvq_dim"], quant_out_dim=q["output_dim"],
        adapter_layers=a["encoder_layers"], adapter_dim=a["d_model"], adapter_heads=a["encoder_attention_heads"],
        adapter_ffn=a["encoder_ffn_dim"], adapter_max_pos=a["max_source_positions"],
        up_stride=gp["upsample_kwargs"]["stride"],
        dec_layers=d["decoder_layers"], dec_dim=d["d_model"], dec_heads=d["decoder_attention_heads"],
        dec_ffn=d["decoder_ffn_dim"],
        dec_max_pos=(d["max_audio_seconds"] * d["sampling_rate"] // d["hop_length"]) // d["stride_size"],
        mel_bins=d["num_mel_bins"], voc_dim=v["dim"], voc_inter=v["intermediate_dim"], voc_layers=v["num_layers"],
        n_fft=v["n_fft"], hop=v["hop_size"],
        mel_n_fft=gp["feature_extractor_kwargs"]["n_fft"], mel_hop=gp["feature_extractor_kwargs"]["hop_length"],
        mel_frames=gp["feature_extractor_kwargs"]["nb_max_frames"],
        enc_layers=gp["semantic_encoder_kwargs"]["encoder_layers"], enc_dim=gp["semantic_encoder_kwargs"]["d_model"],
        enc_heads=gp["semantic_encoder_kwargs"]["encoder_attention_heads"], enc_ffn=gp["semantic_encoder_kwargs"]["encoder_ffn_dim"],
        enc_max_pos=(gp["semantic_encoder_kwargs"]["max_audio_seconds"] * gp["semantic_encoder_kwargs"]["sampling_rate"]
                     // gp["semantic_encoder_kwargs"]["hop_length"]) // gp["semantic_encoder_kwargs"]["stride_size"],
        sem_adapter_layers=gp["semantic_encoder_adapter_kwargs"]["encoder_layers"],
        pre_rvq_layers=gp["pre_rvq_adapter_kwargs"]["encoder_layers"], down_pool=gp["downsample_kwargs"]["avg_pooler"])


def _tlayer(prefix, d, ffn):
    p = prefix
    return [(p + "self_attn.k_proj.weight", (d, d), "lin"), (p + "self_attn.v_proj.weight", (d, d), "lin"),
            (p + "self_attn.v_proj.bias", (d,), "bias"), (p + "self_attn.q_proj.weight", (d, d), "lin"),
            (p + "self_attn.q_proj.bias", (d,), "bias"), (p + "self_attn.out_proj.weight", (d, d), "lin"),
            (p + "self_attn.out_proj.bias", (d,), "bias"), (p + "self_attn_layer_norm.weight", (d,), "norm"),
            (p + "self_attn_layer_norm.bias", (d,), "bias"), (p + "fc1.weight", (ffn, d), "lin"),
            (p + "fc1.bias", (ffn,), "bias"), (p + "fc2.weight", (d, ffn), "lin"), (p + "fc2.bias", (d,), "bias"),
            (p + "final_layer_norm.weight", (d,), "norm"), (p + "final_layer_norm.bias", (d,), "bias")]


def weight_shapes(cfg):
    c = cfg
    out = [("quantizer.output_proj.bias", (c["quant_out_dim"],), "bias"),
           ("quantizer.output_proj.weight_g", (c["quant_out_dim"], 1, 1), "g"),
           ("quantizer.output_proj.weight_v", (c["quant_out_dim"], c["rvq_dim"], 1), "lin")]
    for q in range(c["nq"]):
        out.append((f"quantizer.quantizers.{q}.codebook", (c["codebook_size"], c["rvq_dim"]), "code"))
    d = c["adapter_dim"]
    out += [("post_rvq_adapter.proj.weight", (d, c["quant_out_dim"]), "lin"), ("post_rvq_adapter.proj.bias", (d,), "bias")]
    for n in range(c["adapter_layers"]):
        out += _tlayer(f"post_rvq_adapter.layers.{n}.", d, c["adapter_ffn"])
    out += [("post_rvq_adapter.layer_norm.weight", (d,), "norm"), ("post_rvq_adapter.layer_norm.bias", (d,), "bias"),
            ("post_rvq_adapter.out_proj.weight", (c["quant_out_dim"], d), "lin"),
            ("post_rvq_adapter.out_proj.bias", (c["quant_out_dim"],), "bias"),
            ("upsample.up_conv.weight", (c["up_stride"] * c["dec_dim"], c["dec_dim"], c["up_stride"]), "lin")]
    dd = c["dec_dim"]
    out += [("acoustic_decoder.deconv1.weight", (dd, dd, 3), "lin"), ("acoustic_decoder.deconv1.bias", (dd,), "bias"),
            ("acoustic_decoder.deconv2.weight", (dd, c["mel_bins"], 3), "lin"),
            ("acoustic_decoder.deconv2.bias", (c["mel_bins"],), "bias")]
    for n in range(c["dec_layers"]):
        out += _tlayer(f"acoustic_decoder.layers.{n}.", dd, c["dec_ffn"])
    out += [("acoustic_decoder.layer_norm.weight", (dd,), "norm"), ("acoustic_decoder.layer_norm.bias", (dd,), "bias")]
    v, vi = c["voc_dim"], c["voc_inter"]
    out += [("enhanced_vocos.backbone.embed.weight", (v, c["mel_bins"], 7), "lin"),
            ("enhanced_vocos.backbone.embed.bias", (v,), "bias"),
            ("enhanced_vocos.backbone.norm.weight", (v,), "norm"), ("enhanced_vocos.backbone.norm.bias", (v,), "bias")]
    for n in range(c["voc_layers"]):
        p = f"enhanced_vocos.backbone.convnext.{n}."
        out += [(p + "gamma", (v,), "gamma"), (p + "dwconv.weight", (v, 1, 7), "dw"), (p + "dwconv.bias", (v,), "bias"),
                (p + "norm.weight", (v,), "norm"), (p + "norm.bias", (v,), "bias"),
                (p + "pwconv1.weight", (vi, v), "lin"), (p + "pwconv1.bias", (vi,), "bias"),
                (p + "pwconv2.weight", (v, vi), "lin"), (p + "pwconv2.bias", (v,), "bias")]
    out += [("enhanced_vocos.backbone.final_layer_norm.weight", (v,), "norm"),
            ("enhanced_vocos.backbone.final_layer_norm.bias", (v,), "bias"),
            ("enhanced_vocos.head.out.weight", (c["n_fft"] + 2, v), "lin"),
            ("enhanced_vocos.head.out.bias", (c["n_fft"] + 2,), "bias")]
    return out


def encoder_weight_shapes(cfg):
    """Encode-side tensors (reference state-dict names)."""
    c = cfg
    d, mel = c["enc_dim"], c["mel_bins"]
    out = []
    for enc in ("semantic_encoder", "acoustic_encoder"):
        out += [(f"{enc}.conv1.weight", (d, mel, 3), "lin"), (f"{enc}.conv1.bias", (d,), "bias"),
                (f"{enc}.conv2.weight", (d, d, 3), "lin"), (f"{enc}.conv2.bias", (d,), "bias")]
        for n in range(c["enc_layers"]):
            out += _tlayer(f"{enc}.layers.{n}.", d, c["enc_ffn"])
        out += [(f"{enc}.layer_norm.weight", (d,), "norm"), (f"{enc}.layer_norm.bias", (d,), "bias")]
    for n in range(c["sem_adapter_layers"]):
        out += _tlayer(f"semantic_encoder_adapter.layers.{n}.", d, c["enc_ffn"])
    out += [("semantic_encoder_adapter.layer_norm.weight", (d,), "norm"), ("semantic_encoder_adapter.layer_norm.bias", (d,), "bias"),
            ("pre_rvq_adapter.proj.weight", (d, 2 * d), "lin"), ("pre_rvq_adapter.proj.bias", (d,), "bias")]
    for n in range(c["pre_rvq_layers"]):
        out += _tlayer(f"pre_rvq_adapter.layers.{n}.", d, c["enc_ffn"])
    out += [("pre_rvq_adapter.layer_norm.weight", (d,), "norm"), ("pre_rvq_adapter.layer_norm.bias", (d,), "bias")]
    di = d * c["down_pool"]
    out += [("downsample.gate_proj.weight", (di, d, c["down_pool"]), "lin"), ("downsample.up_proj.weight", (di, d, c["down_pool"]), "lin"),
            ("downsample.down_proj.weight", (di, di), "lin"), ("downsample.layer_norm.weight", (di,), "norm"),
            ("downsample.layer_norm.bias", (di,), "bias"),
            ("quantizer.input_proj.bias", (c["rvq_dim"],), "bias"), ("quantizer.input_proj.weight_g", (c["rvq_dim"], 1, 1), "g"),
            ("quantizer.input_proj.weight_v", (c["rvq_dim"], di, 1), "lin")]
    return out


def synth_weights(cfg, seed, encoder=False):
    """Decode-side tensors; with encoder=True the encode-side ones are appended (drawn AFTER the
    decode-side ones, so decode fixtures do not depend on the flag)."""
    rng = np.random.default_rng(seed)
    w = {}
    shapes = weight_shapes(cfg) + (encoder_weight_shapes(cfg) if encoder else [])
    for name, shape, kind in shapes:
        if kind == "norm":
            a = 1.0 + 0.1 * rng.standard_normal(shape, dtype=np.float32)
        elif kind == "bias":
            a = 0.05 * rng.standard_normal(shape, dtype=np.float32)
        elif kind == "g":
            a = 0.5 + 0.1 * rng.standard_normal(shape, dtype=np.float32)
        elif kind == "gamma":
            a = 0.3 + 0.05 * rng.standard_normal(shape, dtype=np.float32)
        elif kind == "code":
            a = 0.5 * rng.standard_normal(shape, dtype=np.float32)
        elif kind == "dw":
            a = 0.3 * rng.standard_normal(shape, dtype=np.float32)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            a = (1.0 / np.sqrt(fan_in)) * rng.standard_normal(shape, dtype=np.float32)
        w[name] = a.astype(np.float32)
    return w


def synth_codes(cfg, seed, lengths):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, cfg["codebook_size"], (cfg["nq"], int(n))).astype(np.int64) for n in lengths]


def synth_wavs(seed, lengths, sr=16000):
    """Band-limited noise bursts with a slow envelope: gives the mel front-end something speech-like."""
    rng = np.random.default_rng(seed)
    out = []
    for n in lengths:
        t = np.arange(int(n)) / sr
        x = rng.standard_normal(int(n)).astype(np.float32)
        x = np.convolve(x, np.hanning(9).astype(np.float32) / 4.5, mode="same")
        env = 0.5 + 0.5 * np.sin(2 * np.pi * (1.3 + rng.random()) * t + rng.random() * 6.28)
        tone = 0.3 * np.sin(2 * np.pi * (180 + 120 * rng.random()) * t)
        out.append((0.2 * x * env + tone * env).astype(np.float32))
    return out
