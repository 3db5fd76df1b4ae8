"""IO formats around the hot path: PCM wav reading / writing, prompt-audio loading and speaker
merging (reference generation_utils.py:90-177, inference.py:107-111), codec checkpoint loading
(XY_Tokenizer.load_from_checkpoint, reference xy_tokenizer/model.py:28-36).  CPU only.

The wav files are the reference's own examples (read where they lie; the tests skip without /root/reference).
`_resample` stands in for torchaudio.functional.resample, which is not importable here and for which the reference
holds no fixture: PARITY UNPINNED -- the tests below check its length rule and signal properties only."""
import math
import os

import numpy as np
import pytest
import torch

EX = "/root/reference/examples"
needs_examples = pytest.mark.skipif(not os.path.isdir(EX), reason="reference examples not present on this box")


@pytest.fixture(scope="module")
def gu():
    import generation_utils
    return generation_utils


@needs_examples
@pytest.mark.parametrize("name,sr", [("m1.wav", 24000), ("pod_f_enhanced.wav", 16000), ("zh_spk1_moon.wav", 24000)])
def test_read_wav_matches_an_independent_reader(gu, name, sr):
    from scipy.io import wavfile
    wav, got_sr = gu._read_wav(os.path.join(EX, name))
    ref_sr, ref = wavfile.read(os.path.join(EX, name))
    assert got_sr == ref_sr == sr
    assert wav.dtype == torch.float32 and wav.shape == (1, ref.shape[0])
    assert np.array_equal(wav[0].numpy(), ref.astype(np.float32) / 32768.0)


@needs_examples
def test_load_audio_data_single_and_speaker_pair(gu):
    p16, p24a, p24b = (os.path.join(EX, n) for n in ("single_reference.wav", "zh_spk1_moon.wav", "zh_spk2_moon.wav"))
    a = gu.load_audio_data(p16)                                   # already 16 kHz mono: untouched
    w, _ = gu._read_wav(p16)
    assert torch.equal(a, w)
    b = gu.load_audio_data({"speaker1": p24a, "speaker2": p24b})  # resampled 24 k -> 16 k, concatenated in time
    n1, n2 = (gu._read_wav(p)[0].shape[1] for p in (p24a, p24b))
    assert b.shape == (1, math.ceil(n1 * 2 / 3) + math.ceil(n2 * 2 / 3))
    assert torch.isfinite(b).all() and float(b.abs().max()) <= 1.05
    c = gu.load_audio_data((w, 16000))                            # (tensor, sr) pairs pass through (gradio path)
    assert torch.equal(c, w)
    assert gu.load_audio_data(None) is None
    with pytest.raises(ValueError):
        gu.load_audio_data(1234)


def test_merge_speaker_audios_mono_mix_and_rates(gu):
    t = torch.arange(16000, dtype=torch.float32) / 16000.0
    s1 = torch.stack([torch.sin(2 * math.pi * 440 * t), torch.zeros_like(t)])       # stereo: averaged to mono
    s2 = torch.sin(2 * math.pi * 880 * torch.arange(24000, dtype=torch.float32) / 24000.0)[None]
    m = gu.merge_speaker_audios(s1, 16000, s2, 24000)
    assert m.shape == (1, 32000)
    assert torch.allclose(m[0, :16000], 0.5 * s1[0], atol=1e-6)
    # the resampled half is still an 880 Hz tone of the same amplitude (edges aside)
    seg = m[0, 16000 + 800:32000 - 800].double().numpy()
    k = np.arange(seg.size) / 16000.0
    amp = 2 * np.abs(np.mean(seg * np.exp(-2j * math.pi * 880 * k)))
    assert abs(amp - 1.0) < 0.01


def test_resample_length_rule_and_gain(gu):
    """PARITY UNPINNED vs torchaudio (absent): length = ceil(n * new / old), unit DC gain, tone amplitude kept."""
    for sr, tgt, n in ((24000, 16000, 24001), (44100, 16000, 10000), (8000, 16000, 777)):
        x = torch.ones(1, n)
        y = gu._resample(x, sr, tgt)
        assert y.shape == (1, math.ceil(n * tgt / sr))
        mid = y[0, y.shape[1] // 4: 3 * y.shape[1] // 4]
        assert float((mid - 1.0).abs().max()) < 2e-3


def test_save_wav_round_trip(gu, tmp_path):
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.uniform(-1.2, 1.2, (1, 4801)).astype(np.float32))     # beyond [-1,1]: clipped, as PCM16 must
    p = str(tmp_path / "out.wav")
    gu.save_wav(p, x, 24000)
    y, sr = gu._read_wav(p)
    assert sr == 24000 and y.shape == x.shape
    assert float((y - x.clamp(-1, 1)).abs().max()) <= 1.6 / 32768.0        # half an LSB + the 32767/32768 scale pair


def test_codec_checkpoint_round_trip(tmp_path):
    """XY_Tokenizer.load_from_checkpoint reads the yaml's generator_params and the checkpoint's "generator" state dict
    (reference model.py:28-36) with a loader that executes nothing from the file (weights_only)."""
    import yaml
    from mtts import synth_codec
    from XY_Tokenizer.xy_tokenizer.model import XY_Tokenizer
    cfg = synth_codec.reduced(dec_layers=1, voc_layers=1)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth_codec.synth_weights(cfg, 3).items()}
    enc = {"num_mel_bins": 80, "sampling_rate": 16000, "hop_length": 160, "stride_size": 2, "kernel_size": 3, "d_model": 768,
           "encoder_layers": 1, "encoder_attention_heads": 12, "encoder_ffn_dim": 3072, "max_audio_seconds": 30}
    gp = {"input_sample_rate": 16000, "output_sample_rate": 24000, "encoder_downsample_rate": 1280, "decoder_upsample_rate": 1920,
          "feature_extractor_kwargs": {"chunk_length": 30, "feature_size": 80, "hop_length": 160, "n_fft": 400, "sampling_rate": 16000,
                                       "nb_max_frames": 3000},
          "semantic_encoder_kwargs": enc, "acoustic_encoder_kwargs": enc,
          "semantic_encoder_adapter_kwargs": {"encoder_layers": cfg["sem_adapter_layers"]},
          "pre_rvq_adapter_kwargs": {"encoder_layers": cfg["pre_rvq_layers"]}, "downsample_kwargs": {"avg_pooler": 4},
          "quantizer_kwargs": {"num_quantizers": 8, "codebook_size": 1024, "rvq_dim": 512, "output_dim": 3072},
          "post_rvq_adapter_kwargs": {"encoder_layers": cfg["adapter_layers"], "d_model": 768, "encoder_attention_heads": 12,
                                      "encoder_ffn_dim": 3072, "max_source_positions": 375},
          "upsample_kwargs": {"stride": 4},
          "acoustic_decoder_kwargs": {"decoder_layers": cfg["dec_layers"], "d_model": 768, "decoder_attention_heads": 12,
                                      "decoder_ffn_dim": 3072, "max_audio_seconds": 30, "sampling_rate": 16000,
                                      "hop_length": 160, "stride_size": 2, "num_mel_bins": 80},
          "vocos_kwargs": {"dim": 512, "intermediate_dim": 4096, "num_layers": cfg["voc_layers"], "n_fft": 960, "hop_size": 240}}
    cp, yp = str(tmp_path / "xy.ckpt"), str(tmp_path / "xy.yaml")
    torch.save({"generator": sd, "step": 7}, cp)
    with open(yp, "w") as f:
        yaml.safe_dump({"generator_params": gp}, f)
    spt = XY_Tokenizer.load_from_checkpoint(config_path=yp, ckpt_path=cp)
    assert spt.output_sample_rate == 24000 and spt.input_sample_rate == 16000 and spt.nq == 8
    assert spt.cfg["dec_layers"] == 1 and spt.cfg["voc_layers"] == 1
    assert set(spt._sd) == set(sd) and all(torch.equal(spt._sd[k], sd[k]) for k in sd)
    torch.save(sd, cp)                                             # a bare state dict is accepted too
    assert set(XY_Tokenizer.load_from_checkpoint(yp, cp)._sd) == set(sd)
    with pytest.raises(RuntimeError):
        spt.eval().to("cpu").decode([torch.zeros(8, 4, dtype=torch.long)])     # no CPU fallback
