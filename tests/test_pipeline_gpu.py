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


from test_pipeline_gpu_helpers import generator_params as _gp  # noqa: E402


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
    # the three dtypes inference.py offers load; anything else is refused loudly
    m16 = AsteroidTTSInstruct.from_pretrained(str(d), torch_dtype=torch.float16).eval().to("cuda")
    assert m16.dtype == "fp16"
    o16 = m16.generate(input_ids=torch.from_numpy(ids[:3]), attention_mask=torch.from_numpy(mask[:3]))
    assert o16.shape[0] == 3 and o16.shape[1] >= T - 7 + 7
    assert torch.equal(o16[:, :T - 7], torch.from_numpy(ids[:3, :T - 7]))
    with pytest.raises(NotImplementedError):
        AsteroidTTSInstruct.from_pretrained(str(d), torch_dtype=torch.float64)


def test_config5_voice_clone_long_context_end_to_end():
    """BASELINE configs[4] in one piece at reduced width: two dialogues, each with 5 s of prompt audio ->
    spt.encode (HIP encoder) -> prompt layout + delay pattern -> ragged prefill of ~8.2 k / ~7 k tokens -> 90 decode
    steps at an 8.3 k-token context -> un-shift -> decode_each (HIP codec).  Checked against the oracles end to end:
    the prompt codes (encoder oracle, exact), every generated token (the engine's run teacher-forced through the AR
    oracle: the oracle's own decision must be the engine's token wherever its top-2 margin is not a bf16 tie), and
    both waveforms (codec oracle, RMS <= 1e-4)."""
    import generation_utils as gu
    from modeling_asteroid import AsteroidTTSInstruct, GenerationConfig
    from XY_Tokenizer.xy_tokenizer.model import XY_Tokenizer
    cfg = synth.tiny(max_position_embeddings=16384)
    w = synth.synth_weights(cfg, 301, emb_row_sigma=0.6, speech_boost=6.0, eos_boost=1.0)
    new = 90
    model = AsteroidTTSInstruct.from_state_dict(cfg, w, GenerationConfig(max_new_tokens=new, eos_token_id=cfg["eos_token_id"]))
    ccfg = synth_codec.reduced(dec_layers=1, voc_layers=1, enc_layers=1)
    cw = synth_codec.synth_weights(ccfg, 12, encoder=True)
    spt = XY_Tokenizer(_gp(ccfg), cw).eval().to("cuda")
    model = model.eval().to("cuda")
    wavs = [torch.from_numpy(x)[None] for x in synth_codec.synth_wavs(5, [16000 * 5, 16000 * 5 - 700])]
    rng = np.random.default_rng(4)
    long_text = ["[S1]" + "".join(chr(int(c)) for c in rng.integers(0x4e00, 0x9000, n)) for n in (8100, 6900)]
    items = [{"text": long_text[0], "prompt_audio": (wavs[0], 16000), "prompt_text": "[S1]reference one"},
             {"text": long_text[1], "prompt_audio": (wavs[1], 16000), "prompt_text": "[S2]reference two"}]
    texts, results = gu.process_batch(items, Tok(), model, spt, "cuda", "sys", 0)
    assert all(r is not None for r in results)
    # (1) prompt codes: the engine was fed exactly what the encoder oracle produces
    enc = co.CodecEncodeOracle({**synth_codec.codec_config(), **ccfg}, cw)
    for wv in wavs:
        want = enc.encode([wv[0].numpy()])[0]
        got = spt.encode([wv[0]])["codes_list"][0].cpu().numpy()
        assert got.shape == (8, wv.shape[1] // 1280) and np.array_equal(got, want)
    # (2) the same prompts through the glue, the engine's tokens through the AR oracle
    seqs = [gu.shifting_inputs(gu.process_inputs(Tok(), spt, "sys", t["final_text"], "cuda", audio_data=wv), Tok())
            for t, wv in zip(texts, wavs)]
    ids, mask = gu.rpadding(seqs, 8, Tok())
    T = ids.shape[1]
    assert T > 8200 and int(mask[1].sum()) < T - 1000           # 8.2 k context, ragged by more than a thousand tokens
    out = model.generate(input_ids=ids.cuda(), attention_mask=mask.cuda()).cpu().numpy()
    assert out.shape[1] >= T - 7 + new
    orc = ao.AsteroidOracle(cfg, w, "bf16")
    orc.prefill_chunk = 512
    _, odec, _ = orc.generate(ids.numpy(), mask.numpy(), T + new, forced=out)
    margins = np.stack(orc.last_margins)
    want = out[:, T - 7:].transpose(1, 0, 2)[:odec.shape[0]]
    free = np.ones_like(margins, dtype=bool)
    for s in range(7):
        free[s, :, s + 1:] = False
    safe = free & (margins >= 0.02)          # oracle-vs-engine gate (tests/test_engine_gpu.py: MARGIN_ORACLE)
    assert safe.sum() > 0.8 * free.sum()
    assert np.array_equal(odec[safe], want[safe])
    assert np.array_equal(odec[~free], want[~free])
    assert int(model._engine.seq_state()[2].max()) >= 8300     # the decode did run at an 8.3 k context
    # (3) waveforms: un-shift, codec oracle
    speech = ao.unshift_outputs(out, T - 7)
    last = ao.find_max_valid_positions(speech)
    dec = co.CodecOracle(ccfg, cw)
    for i, r in enumerate(results):
        assert last[i] + 1 >= new - 7
        ref = dec.decode([speech[i, :last[i] + 1].T])[0]
        got = r["audio_data"][0].numpy()
        assert got.shape == ref.shape
        assert np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2)) <= 1e-4


def test_more_than_128_rows_take_the_scheduled_path_and_equal_their_batch1_runs():
    """AsteroidTTSInstruct.generate with 134 rows (> MAX_ENGINE_BATCH): `_generate_scheduled` serves them through the
    continuous batcher.  Every row -- sampled on all channels -- equals its own batch-1 run with the same seed and its
    Philox row id (so the same seed and prompts give the same tokens on either side of the 128-row limit), the rows that
    finish early are followed by the reference's finished-row padding (eos, 1024 x 7), and the first 128 rows equal
    ONE static batch of those rows wherever no dialogue is cut off by max_length (the reference keeps evaluating such
    rows, the scheduler lets them leave)."""
    from modeling_asteroid import AsteroidTTSInstruct, GenerationConfig
    from mtts.engine import Engine
    cfg = synth.tiny()
    w = synth.synth_weights(cfg, 311, emb_row_sigma=0.6, speech_boost=5.0, eos_boost=5.0)
    layers = [dict(top_k=20, top_p=0.9, temperature=1.1, repetition_penalty=1.1)] * 8
    gc = GenerationConfig(max_new_tokens=30, do_samples=[True] * 8, layers=layers, eos_token_id=cfg["eos_token_id"])
    model = AsteroidTTSInstruct.from_state_dict(cfg, w, gc).eval().to("cuda")
    B = 134
    ids, mask = synth.synth_prompts(cfg, 312, B, 40, 0.3, True)
    T = ids.shape[1]
    out = model.generate(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), seed=77).numpy()
    assert out.shape[0] == B and out.shape[2] == 8
    solo = Engine(cfg, max_batch=1, max_seq_len=256)
    solo.bind_state_dict(w)
    eos = cfg["eos_token_id"]
    early = 0
    for b in range(B):
        pad = int(np.argmax(mask[b] > 0))
        p = ids[b, pad:]
        alone = solo.generate(p[None], np.ones((1, p.shape[0])), p.shape[0] + 30, layers=layers, do_samples=[True] * 8,
                              seed=77, row_ids=[b])[0]
        gen = alone[p.shape[0] - 7:]
        assert np.array_equal(out[b, T - 7:T - 7 + gen.shape[0]], gen), b
        rest = out[b, T - 7 + gen.shape[0]:]
        assert (rest[:, 0] == eos).all() and (rest[:, 1:] == 1024).all(), b
        early += int(gen.shape[0] < out.shape[1] - (T - 7))
    solo.close()
    assert early >= 5                                          # some dialogues did finish early and were padded
    # the first 128 rows as one static batch with the same seed
    static = model.generate(input_ids=torch.from_numpy(ids[:128]), attention_mask=torch.from_numpy(mask[:128]), seed=77).numpy()
    n = min(static.shape[1], out.shape[1])
    same = [b for b in range(128) if np.array_equal(static[b, :n], out[b, :n])]
    assert len(same) >= 120, len(same)


def test_sharded_entry_point_world1_equals_process_batch():
    """inference_sharded.process_batch_sharded without a process group (one GPU) is generation_utils.process_batch:
    same text records, same audio bit for bit (greedy)."""
    import generation_utils as gu
    import inference_sharded as ish
    from modeling_asteroid import AsteroidTTSInstruct, GenerationConfig
    from XY_Tokenizer.xy_tokenizer.model import XY_Tokenizer
    cfg = synth.tiny()
    w = synth.synth_weights(cfg, 103, emb_row_sigma=0.6, speech_boost=4.0, eos_boost=1.0)
    model = AsteroidTTSInstruct.from_state_dict(cfg, w, GenerationConfig(max_new_tokens=20, eos_token_id=cfg["eos_token_id"]))
    ccfg = synth_codec.reduced(dec_layers=1, voc_layers=1)
    spt = XY_Tokenizer(_gp(ccfg), synth_codec.synth_weights(ccfg, 9)).eval().to("cuda")
    model = model.eval().to("cuda")
    items = [{"text": "[S1]One.[S2]Two, a little longer."}, {"text": "[S1]Three"}, {"text": "[S2]And the fourth item."}]
    assert ish.estimate_work(items, Tok(), "sys") == [len(it["text"].replace("[S1]", "<speaker1>").replace("[S2]", "<speaker2>")) for it in items]
    t1, a1 = gu.process_batch(items, Tok(), model, spt, "cuda", "sys", 2)
    t2, a2 = ish.process_batch_sharded(items, Tok(), model, spt, "cuda", "sys", 2)
    assert t1 == t2 and len(a1) == len(a2) == 3
    for x, y in zip(a1, a2):
        assert x["index"] == y["index"] and x["sample_rate"] == y["sample_rate"]
        assert torch.equal(x["audio_data"], y["audio_data"])


def test_finetune_data_path_through_the_hip_encoder(tmp_path):
    """finetune/data_preprocess.py on the real engine: both JSONL formats, `spt.encode` = the HIP
    encoder.  The speech rows of input_ids are the encoder oracle's codes (+151665 on channel 0), two recordings are
    joined at token level, labels follow the reference's masks, and the files have the reference's layout."""
    import pickle
    import generation_utils as gu
    from finetune import data_preprocess as dp
    from XY_Tokenizer.xy_tokenizer.model import XY_Tokenizer
    ccfg = synth_codec.reduced(dec_layers=1, voc_layers=1, enc_layers=1)
    cw = synth_codec.synth_weights(ccfg, 14, encoder=True)
    spt = XY_Tokenizer(_gp(ccfg), cw).eval()
    wavs = synth_codec.synth_wavs(8, [16000 * 3, 16000 * 2 + 500, 16000 * 4])
    for name, x in zip(("a.wav", "ref.wav", "main.wav"), wavs):
        gu.save_wav(str(tmp_path / name), torch.from_numpy(x)[None], 16000)
    items = [{"file_path": str(tmp_path / "a.wav"), "full_transcript": "[S1]Hello there![S2]Hi."},
             {"reference_audio": str(tmp_path / "ref.wav"), "reference_text": "[S1]ref.", "audio": str(tmp_path / "main.wav"), "text": "[S2]main text."}]
    jl = tmp_path / "d.jsonl"
    jl.write_text("\n".join(json.dumps(it) for it in items) + "\n")

    class Tok2(Tok):
        def encode(self, s, add_special_tokens=True):
            return [min(ord(c), 151000) for c in s]

    dp.process_data(str(jl), "unused", str(tmp_path / "out"), data_name="d", use_normalize=True, tokenizer=Tok2(), spt=spt, device="cuda")
    metas = np.load(tmp_path / "out" / "d_metas.npy")
    assert metas.shape == (3, 2)
    enc = co.CodecEncodeOracle({**synth_codec.codec_config(), **ccfg}, cw)
    # what the files' audio was after the PCM16 round trip of save_wav / _read_wav
    loaded = [gu.load_audio_data(str(tmp_path / n))[0].numpy() for n in ("a.wav", "ref.wav", "main.wav")]
    want_codes = [enc.encode([x])[0].T for x in loaded]                      # [frames, 8]
    want_audio = [want_codes[0], np.concatenate([want_codes[1], want_codes[2]], axis=0)]
    with open(tmp_path / "out" / "d.pkl", "rb") as f:                         # written by this test
        for k, off in enumerate(metas[0]):
            f.seek(int(off))
            e = pickle.load(f)
            ids, labels = np.array(e["input_ids"]), np.array(e["labels"])
            total, n_audio = int(metas[1, k]), int(metas[2, k])
            assert ids.shape == labels.shape == (total, 8) and n_audio == want_audio[k].shape[0]
            a0 = total - n_audio - len("<|end_of_speech|>")
            speech = ids[a0:a0 + n_audio].copy()
            speech[:, 0] -= 151665
            assert np.array_equal(speech, want_audio[k]), k
            assert np.array_equal(labels[a0:a0 + n_audio], ids[a0:a0 + n_audio]) and (labels[:a0] == -100).all()
            assert np.array_equal(labels[a0 + n_audio:, 0], ids[a0 + n_audio:, 0])
    assert list(metas[2]) == [37, 25 + 50]


def _two_rank_worker(rank, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        import sys
        here = os.path.dirname(os.path.abspath(__file__))
        for pth in (os.path.dirname(here), os.path.join(os.path.dirname(here), "moss-ttsd_amd"), here):
            if pth not in sys.path:
                sys.path.insert(0, pth)
        import torch.distributed as dist
        import inference_sharded as ish
        world, r, dev = ish.init_distributed(backend="gloo")
        torch.manual_seed(123)
        tok, model, spt = ish.load_model_sharded("m", "c", "k", device=dev, loader=_tiny_loader)
        texts, audio = ish.process_batch_sharded(_SHARD_ITEMS, tok, model, spt, str(dev), "sys", 10)
        if r == 0:
            q.put(("ok", texts, [None if a is None else a["audio_data"].numpy() for a in audio]))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put(("err", repr(e) + traceback.format_exc(), None))


_SHARD_ITEMS = [{"text": "[S1]" + "word " * (3 + 5 * (i % 4)) + "[S2]ok."} for i in range(7)]


def _tiny_loader(model_path, spt_cfg, spt_ckpt, torch_dtype=None, attn_implementation=None):
    from modeling_asteroid import AsteroidTTSInstruct, GenerationConfig
    from XY_Tokenizer.xy_tokenizer.model import XY_Tokenizer
    cfg = synth.tiny()
    w = synth.synth_weights(cfg, 321, emb_row_sigma=0.6, speech_boost=5.0, eos_boost=4.0)
    layers = [dict(top_k=20, top_p=0.9, temperature=1.05, repetition_penalty=1.1)] * 8
    gc = GenerationConfig(max_new_tokens=26, do_samples=[True] * 8, layers=layers, eos_token_id=cfg["eos_token_id"])
    model = AsteroidTTSInstruct.from_state_dict(cfg, {k: torch.from_numpy(v).to(torch.bfloat16) for k, v in w.items()}, gc)
    ccfg = synth_codec.reduced(dec_layers=1, voc_layers=1)
    spt = XY_Tokenizer(_gp(ccfg), {k: torch.from_numpy(v) for k, v in synth_codec.synth_weights(ccfg, 9).items()})
    return Tok(), model.eval(), spt.eval()


def test_sharded_two_ranks_equal_the_single_process_run():
    """BASELINE configs[3]'s flow on hardware, minus RCCL: two ranks (gloo; both on this box's one GPU) run
    inference_sharded -- rank 0 alone builds the models, weights travel as flat buckets, the 7 items are dealt by
    estimated work, each rank runs process_batch on its share with its rows' job-wide Philox ids, audio and text records
    meet on rank 0.  SAMPLED on all 8 channels: the result must equal one process_batch over the whole batch in this
    process (same seed), token for token and therefore sample for sample."""
    import socket
    import torch.multiprocessing as mp
    import generation_utils as gu
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    status, texts, audio = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
    assert status == "ok", texts
    torch.manual_seed(123)
    tok, model, spt = _tiny_loader("m", "c", "k")
    want_t, want_a = gu.process_batch(_SHARD_ITEMS, tok, model.to("cuda"), spt.to("cuda"), "cuda", "sys", 10)
    assert texts == want_t
    assert len(audio) == len(want_a) == 7
    for got, ref in zip(audio, want_a):
        assert (got is None) == (ref is None)
        if ref is not None:
            assert got.shape == tuple(ref["audio_data"].shape)
            assert np.array_equal(got, ref["audio_data"].numpy())
    assert not np.array_equal(audio[0], audio[1])          # (different dialogues, different draws)
