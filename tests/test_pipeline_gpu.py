"""Drop-in surface end to end on the GPU: generation_utils.process_batch with the MI355X
model + codec (config 1 plumbing with a stub tokenizer; there is no real checkpoint here)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from mtts import synth, synth_codec  # noqa: E402
from oracle import asteroid_oracle as ao  # noqa: E402
from oracle import codec_oracle as co  # noqa: E402


class Tok:
    pad_token_id = 151643

    def encode(self, s):
        return [min(ord(c), 151000) for c in s]


def _gp(cfg):
    """generator_params dict (yaml shape) for a reduced-depth codec."""
    enc = {"encoder_layers": cfg["enc_layers"], "d_model": 768, "encoder_attention_heads": 12, "encoder_ffn_dim": 3072,
           "max_audio_seconds": 30, "sampling_rate": 16000, "hop_length": 160, "stride_size": 2}
    return {"input_sample_rate": 16000, "output_sample_rate": 24000,
            "feature_extractor_kwargs": {"n_fft": 400, "hop_length": 160, "nb_max_frames": 3000},
            "semantic_encoder_kwargs": enc, "acoustic_encoder_kwargs": enc,
            "semantic_encoder_adapter_kwargs": {"encoder_layers": cfg["sem_adapter_layers"]},
            "pre_rvq_adapter_kwargs": {"encoder_layers": cfg["pre_rvq_layers"]},
            "downsample_kwargs": {"avg_pooler": 4},
            "quantizer_kwargs": {"num_quantizers": 8, "codebook_size": 1024, "rvq_dim": 512, "output_dim": 3072},
            "post_rvq_adapter_kwargs": {"encoder_layers": cfg["adapter_layers"], "d_model": 768,
                                        "encoder_attention_heads": 12, "encoder_ffn_dim": 3072, "max_source_positions": 375},
            "upsample_kwargs": {"stride": 4},
            "acoustic_decoder_kwargs": {"decoder_layers": cfg["dec_layers"], "d_model": 768, "decoder_attention_heads": 12,
                                        "decoder_ffn_dim": 3072, "max_audio_seconds": 30, "sampling_rate": 16000,
                                        "hop_length": 160, "stride_size": 2, "num_mel_bins": 80},
            "vocos_kwargs": {"dim": 512, "intermediate_dim": 4096, "num_layers": cfg["voc_layers"], "n_fft": 960, "hop_size": 240}}


def test_process_batch_text_only_end_to_end():
    import generation_utils as gu
    from modeling_asteroid import AsteroidTTSInstruct, GenerationConfig
    from XY_Tokenizer.xy_tokenizer.model import XY_Tokenizer
    cfg = synth.tiny()
    w = synth.synth_weights(cfg, 103, emb_row_sigma=0.6, speech_boost=4.0, eos_boost=1.0)
    model = AsteroidTTSInstruct.from_state_dict(cfg, w, GenerationConfig(max_new_tokens=24, eos_token_id=cfg["eos_token_id"]))
    ccfg = synth_codec.reduced(dec_layers=1, voc_layers=2)
    cw = synth_codec.synth_weights(ccfg, 9)
    spt = XY_Tokenizer(_gp(ccfg), cw)
    model = model.eval().to("cuda")
    spt = spt.eval().to("cuda")
    items = [{"text": "[S1]Hello there.[S2]Hi!"}, {"text": "[S1]A much longer line of dialogue for the second item."}]
    texts, results = gu.process_batch(items, Tok(), model, spt, "cuda", "You are a speech synthesizer.", 5, use_normalize=True)
    assert [t["index"] for t in texts] == [5, 6]
    assert texts[0]["final_text"].startswith("<speaker1>") and texts[0]["normalized_text"] is not None
    assert len(results) == 2
    for r in results:
        assert r is not None and r["sample_rate"] == 24000
        a = r["audio_data"]
        assert a.dim() == 2 and a.shape[0] == 1 and a.shape[1] % 1920 == 0 and a.shape[1] > 0
        assert a.device.type == "cpu" and torch.isfinite(a).all()
    # the same ids through the oracle's glue + codec oracle give the same waveform
    seqs = [gu.shifting_inputs(gu.process_inputs(Tok(), spt, "You are a speech synthesizer.", t["final_text"], "cuda"), Tok())
            for t in texts]
    ids, mask = gu.rpadding(seqs, 8, Tok())
    out = model.generate(input_ids=ids.cuda(), attention_mask=mask.cuda()).cpu().numpy()
    T = ids.shape[1]
    speech = ao.unshift_outputs(out, T - 7)
    last = ao.find_max_valid_positions(speech)
    orc = co.CodecOracle(ccfg, cw)
    for i, r in enumerate(results):
        want = orc.decode([speech[i, :last[i] + 1].T])[0]
        got = r["audio_data"][0].numpy()
        assert got.shape == want.shape
        assert np.sqrt(np.mean((got.astype(np.float64) - want) ** 2)) <= 1e-4
    # a sample whose codec decode fails comes back as None, the others survive
    class Broken:
        output_sample_rate = 24000
        def decode(self, codes_list, overlap_seconds=10):
            if codes_list[0].shape[-1] == int(last[0]) + 1:
                raise RuntimeError("boom")
            return spt.decode(codes_list, overlap_seconds)
    if int(last[0]) != int(last[1]):
        _, res2 = gu.process_batch(items, Tok(), model, Broken(), "cuda", "You are a speech synthesizer.", 0)
        assert res2[0] is None and res2[1] is not None


def test_process_batch_voice_clone_prompt():
    """Prompt audio goes through spt.encode (HIP encoder), its codes are teacher-forced through the
    delay pattern, and the batch mixes a cloned and a text-only item (ragged left padding)."""
    import generation_utils as gu
    from modeling_asteroid import AsteroidTTSInstruct, GenerationConfig
    from XY_Tokenizer.xy_tokenizer.model import XY_Tokenizer
    cfg = synth.tiny()
    w = synth.synth_weights(cfg, 104, emb_row_sigma=0.6, speech_boost=4.0, eos_boost=1.0)
    model = AsteroidTTSInstruct.from_state_dict(cfg, w, GenerationConfig(max_new_tokens=16, eos_token_id=cfg["eos_token_id"]))
    ccfg = synth_codec.reduced(dec_layers=1, voc_layers=1, enc_layers=1)
    cw = synth_codec.synth_weights(ccfg, 10, encoder=True)
    spt = XY_Tokenizer(_gp(ccfg), cw).eval().to("cuda")
    model = model.eval().to("cuda")
    wav = torch.from_numpy(synth_codec.synth_wavs(3, [16000 * 2])[0])[None]
    items = [{"text": "[S1]Cloned voice line.", "prompt_audio": (wav, 16000), "prompt_text": "[S1]reference words"},
             {"text": "[S2]Plain item."}]
    texts, results = gu.process_batch(items, Tok(), model, spt, "cuda", "sys", 0)
    assert texts[0]["original_text"] == "[S1]reference words[S1]Cloned voice line."
    assert all(r is not None and r["audio_data"].shape[1] % 1920 == 0 for r in results)
    # the prompt codes the engine was fed are exactly what the codec oracle's encoder produces
    orc = co.CodecEncodeOracle({**synth_codec.codec_config(), **ccfg}, cw)
    want = orc.encode([wav[0].numpy()])[0]
    got = spt.encode([wav[0]])["codes_list"][0].cpu().numpy()
    assert got.shape == (8, 25) and np.array_equal(got, want)


def test_model_requires_gpu_and_bf16():
    from modeling_asteroid import AsteroidTTSInstruct
    cfg = synth.tiny()
    m = AsteroidTTSInstruct.from_state_dict(cfg, {})
    with pytest.raises(RuntimeError):
        m.generate(input_ids=torch.zeros(1, 9, 8, dtype=torch.long), attention_mask=torch.ones(1, 9))


def test_overlapped_decode_equals_sequential():
    """Streaming windows decoded on a side stream during generation == per-sample decode afterwards."""
    from mtts.engine import Engine
    from mtts.codec import CodecEngine
    from mtts import streaming
    cfg = synth.tiny()
    w = synth.synth_weights(cfg, 7, emb_row_sigma=0.3, speech_boost=8.0, eos_boost=1.0)
    eng = Engine(cfg, max_batch=3, max_seq_len=1024)
    eng.bind_state_dict(w)
    ccfg = synth_codec.reduced(dec_layers=1, voc_layers=1)
    cod = CodecEngine(ccfg)
    cod.bind_state_dict(synth_codec.synth_weights(ccfg, 3))
    ids, mask = synth.synth_prompts(cfg, 5, 3, 24, 0.0, True)
    max_length = ids.shape[1] + 700                     # 700+ frames: two full windows + a tail per row
    layers = [dict(top_k=20, top_p=0.9, temperature=1.0)] * 8
    gen, wavs = streaming.generate_with_overlapped_decode(eng, cod, ids, mask, max_length, layers=layers,
                                                          do_samples=[True] * 8, seed=11)
    lens = streaming.valid_lengths(gen)
    assert (lens > 2 * 250 + 10).all(), lens
    # sequential reference: same tokens (same seed), decode each sample on its own after the loop
    out = eng.generate(ids, mask, max_length, layers=layers, do_samples=[True] * 8, seed=11)
    T = ids.shape[1]
    assert np.array_equal(out[:, T - 7:].transpose(1, 0, 2), gen)
    speech = ao.unshift_outputs(out, T - 7)
    last = ao.find_max_valid_positions(speech)
    for b in range(3):
        assert int(last[b]) + 1 == int(lens[b])
        want = cod.decode([torch.from_numpy(speech[b, :last[b] + 1].T.copy())])[0]
        assert wavs[b].shape == want.shape
        assert torch.equal(wavs[b], want), float((wavs[b] - want).abs().max())
    eng.close()
    cod.close()


def test_from_pretrained_directory_and_large_batch(tmp_path):
    """Local checkpoint directory (config.json + generation_config.json + safetensors) -> from_pretrained;
    a batch larger than the engine's 32 rows is served in slices with the reference's finished-row padding."""
    from safetensors.torch import save_file
    from modeling_asteroid import AsteroidTTSInstruct
    cfg = synth.tiny()
    w = synth.synth_weights(cfg, 55, emb_row_sigma=0.6, speech_boost=3.3, eos_boost=3.3)
    d = tmp_path / "ckpt"
    d.mkdir()
    hf_cfg = dict(cfg)
    hf_cfg["rope_parameters"] = {"rope_theta": cfg["rope_theta"], "rope_type": "default"}
    hf_cfg.pop("rope_theta")
    (d / "config.json").write_text(json.dumps(hf_cfg))
    (d / "generation_config.json").write_text(json.dumps({"max_new_tokens": 12, "eos_token_id": cfg["eos_token_id"],
                                                          "do_samples": [False] * 8, "layers": [{}] * 8}))
    sd = {k: torch.from_numpy(v).to(torch.bfloat16) for k, v in w.items()}
    sd["lm_heads.0.weight"] = sd["model.embedding_list.0.weight"].clone()      # tied duplicates are ignored
    save_file(sd, str(d / "model.safetensors"))
    model = AsteroidTTSInstruct.from_pretrained(str(d), torch_dtype=torch.bfloat16, attn_implementation="sdpa").eval().to("cuda")
    assert model.config.rope_theta == cfg["rope_theta"] and model.config.head_dim == 128
    ids, mask = synth.synth_prompts(cfg, 56, 40, 20, 0.3, True)
    out = model.generate(input_ids=torch.from_numpy(ids).cuda(), attention_mask=torch.from_numpy(mask).cuda())
    assert out.is_cuda and out.shape[0] == 40 and out.shape[2] == 8
    T = ids.shape[1]
    a = model.generate(input_ids=torch.from_numpy(ids[:32]), attention_mask=torch.from_numpy(mask[:32])).numpy()
    b = model.generate(input_ids=torch.from_numpy(ids[32:]), attention_mask=torch.from_numpy(mask[32:])).numpy()
    o = out.cpu().numpy()
    assert np.array_equal(o[:32, :a.shape[1]], a) and np.array_equal(o[32:, :b.shape[1]], b)
    # rows of the shorter slice are padded the way finished rows are: (eos, 1024 x 7)
    if a.shape[1] != b.shape[1]:
        short = slice(0, 32) if a.shape[1] < b.shape[1] else slice(32, 40)
        n = min(a.shape[1], b.shape[1])
        assert (o[short, n:, 0] == cfg["eos_token_id"]).all() and (o[short, n:, 1:] == 1024).all()
    with pytest.raises(NotImplementedError):
        AsteroidTTSInstruct.from_pretrained(str(d), torch_dtype=torch.float16)
