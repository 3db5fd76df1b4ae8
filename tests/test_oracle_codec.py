"""Codec oracle (numpy) vs fixtures produced by the reference XY_Tokenizer.decode (CPU)."""
import json
import os

import numpy as np
import pytest

from mtts import synth_codec
from oracle import codec_oracle as co

CASES = ["codec_T40", "codec_ragged_1win", "codec_T600", "codec_full_T24"]
RMS_TOL = 1e-4      # north_star: waveform RMS within 1e-4 for the codec decoder


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = json.loads(str(z["cfg"]))
    w = synth_codec.synth_weights(cfg, int(z["seed"]))
    codes = synth_codec.synth_codes(cfg, int(z["seed"]) + 1, list(z["lengths"]))
    return z, cfg, w, codes


def check_against_fixture(z, wavs):
    stride = int(z["stride"])
    for i, wv in enumerate(wavs):
        assert wv.shape[0] == int(z[f"wav{i}_len"])
        ref = z[f"wav{i}_sub"]
        err = wv[::stride].astype(np.float64) - ref.astype(np.float64)
        rms_err = float(np.sqrt(np.mean(err ** 2)))
        assert rms_err <= RMS_TOL, (i, rms_err)
        assert float(np.abs(err).max()) <= 1e-3
        sec = 24000
        rms = np.array([np.sqrt(np.mean(wv[s:s + sec].astype(np.float64) ** 2)) for s in range(0, wv.shape[0], sec)])
        np.testing.assert_allclose(rms, z[f"wav{i}_rms"], atol=RMS_TOL)


@pytest.mark.parametrize("name", CASES)
def test_codec_oracle_matches_reference(golden_dir, name):
    z, cfg, w, codes = load(golden_dir, name)
    if name == "codec_T600":
        pytest.importorskip("scipy")
    wavs = co.CodecOracle(cfg, w).decode(codes)
    check_against_fixture(z, wavs)


def test_codec_oracle_edge_cases():
    cfg = synth_codec.reduced(dec_layers=1, voc_layers=1)
    w = synth_codec.synth_weights(cfg, 1)
    orc = co.CodecOracle(cfg, w)
    # window bookkeeping: 251 codes -> 2 windows, output exactly 251*1920 samples
    out = orc.decode(synth_codec.synth_codes(cfg, 2, [251]))
    assert out[0].shape[0] == 251 * 1920
    # one code
    out = orc.decode(synth_codec.synth_codes(cfg, 3, [1]))
    assert out[0].shape[0] == 1920 and np.isfinite(out[0]).all()


ENC_CASES = ["codec_enc_3s", "codec_enc_ragged", "codec_enc_35s"]


@pytest.mark.parametrize("name", ENC_CASES)
def test_codec_encode_oracle_matches_reference(golden_dir, name):
    """Exact code ids against the reference XY_Tokenizer.encode (CPU) on synthetic audio."""
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = json.loads(str(z["cfg"]))
    w = synth_codec.synth_weights(cfg, int(z["seed"]), encoder=True)
    wavs = synth_codec.synth_wavs(int(z["seed"]) + 1, list(z["lengths"]))
    got = co.CodecEncodeOracle(cfg, w).encode(wavs)
    for i, g in enumerate(got):
        want = z[f"codes{i}"].astype(np.int64)
        assert g.shape == want.shape, (g.shape, want.shape)
        assert np.array_equal(g, want), float((g != want).mean())


def test_mel_filter_bank_matches_transformers():
    tr = pytest.importorskip("transformers.audio_utils")
    ref = tr.mel_filter_bank(num_frequency_bins=201, num_mel_filters=80, min_frequency=0.0, max_frequency=8000.0,
                             sampling_rate=16000, norm="slaney", mel_scale="slaney")
    got = co.mel_filter_bank_slaney(201, 80, 16000, 0.0, 8000.0)
    np.testing.assert_allclose(got, ref.astype(np.float32), rtol=1e-6, atol=1e-9)
