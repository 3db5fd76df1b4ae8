"""Host glue (delay pattern, padding, text normalisation, item parsing) -- CPU only."""
import importlib
import sys
import types

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st


@pytest.fixture(scope="module")
def gu():
    # the drop-in modules import the engines lazily; stub nothing, just import
    import generation_utils
    return generation_utils


class Tok:
    pad_token_id = 151643

    def encode(self, s):
        return [min(ord(c), 151000) for c in s]


def test_shift_pad_unshift_roundtrip(gu):
    rng = np.random.default_rng(0)
    raws = [rng.integers(0, 1024, (n, 8)) for n in (5, 11, 8)]
    sh = [gu.shifting_inputs(r, Tok()) for r in raws]
    assert [s.shape[0] for s in sh] == [12, 18, 15]
    ids, mask = gu.rpadding(sh, 8, Tok())
    assert ids.shape == (3, 18, 8) and mask.shape == (3, 18)
    assert mask[0, :6].sum() == 0 and mask[0, 6:].sum() == 12
    assert (ids[0, :6, 0] == 151643).all() and (ids[0, :6, 1:] == 1024).all()
    # un-shift recovers the rows (generation_utils.py:416-425 semantics)
    out = ids[1:2]
    seq_len = out.shape[1] - 7
    rec = torch.stack([out[0, j:seq_len + j, j] for j in range(8)], dim=-1)
    assert np.array_equal(rec.numpy(), raws[1])


def test_find_max_valid_positions(gu):
    c = torch.full((3, 6, 8), 1024)
    c[0, :4, 1] = 7
    c[2, 5, 1] = 3
    c[1, :, 0] = 5            # channel 0 does not count: the reference looks at channel 1
    assert gu.find_max_valid_positions(c).tolist() == [3, -1, 5]


def test_normalize_text_cases(gu):
    f = gu.normalize_text
    assert f("[1]你好！[2]哈哈哈，是吗？") == "[S1]你好。[S2](笑)，是吗。"
    assert f("[S1]Hello: world; ok! [S1]again?") == "[S1]Hello, world, ok.again."
    assert f("no tags 【here】") == "no tags here"
    assert f("[note]x[S2]haha yes……no") == "notex[S2](laughs) yes，no"
    assert f("") == ""


def test_process_jsonl_item_formats(gu, capsys):
    a = gu.process_jsonl_item({"text": "[S1]hi", "prompt_audio": "a.wav", "prompt_text": "[S1]p", "base_path": "/x"})
    assert a == {"text": "[S1]hi", "prompt_text": "[S1]p", "prompt_audio": "/x/a.wav"}
    b = gu.process_jsonl_item({"text": "t", "prompt_audio_speaker1": "s1.wav", "prompt_text_speaker1": "one",
                               "prompt_audio_speaker2": "s2.wav", "prompt_text_speaker2": "two", "base_path": "b"})
    assert b["prompt_audio"] == {"speaker1": "b/s1.wav", "speaker2": "b/s2.wav"}
    assert b["prompt_text"] == "[S1]one[S2]two"
    c = gu.process_jsonl_item({"text": "only"})
    assert c == {"text": "only", "prompt_text": "", "prompt_audio": None}


def test_process_inputs_text_only(gu):
    ids = gu.process_inputs(Tok(), None, "sys", "[S1]x", "cpu")
    assert ids.shape[1] == 8 and (ids[:, 1:] == 1024).all()
    assert ids.shape[0] == len("<|begin_of_style|>sys<|end_of_style|>\n<|begin_of_text|>[S1]x<|end_of_text|>\n<|begin_of_speech|>")


@settings(max_examples=30, deadline=None)
@given(st.lists(st.integers(min_value=1, max_value=20), min_size=1, max_size=5))
def test_left_pad_property(lengths):
    import generation_utils as gu
    rng = np.random.default_rng(sum(lengths))
    sh = [gu.shifting_inputs(rng.integers(0, 1024, (n, 8)), Tok()) for n in lengths]
    ids, mask = gu.rpadding(sh, 8, Tok())
    for b, s in enumerate(sh):
        assert int(mask[b].sum()) == s.shape[0]
        assert np.array_equal(ids[b, ids.shape[1] - s.shape[0]:].numpy(), s)
        assert (mask[b].numpy()[:-1] <= mask[b].numpy()[1:]).all()      # left padded: mask is non-decreasing


def test_generation_config_channel_settings():
    from modeling_asteroid import GenerationConfig, AsteroidTTSConfig
    g = GenerationConfig(do_samples=[True] * 8, layers=[{"top_k": 50}] * 8)
    layers, ds = g.channel_settings(8)
    assert ds == [True] * 8 and layers[3] == {"top_k": 50}
    g = GenerationConfig(do_sample=True, temperature=0.8, top_p=0.9)
    layers, ds = g.channel_settings(8)
    assert ds == [True] * 8 and layers[0]["temperature"] == 0.8 and layers[0]["top_k"] is None
    c = AsteroidTTSConfig(hidden_size=256, num_attention_heads=2, speech_token_range=[10, 20], head_dim=None)
    assert c.head_dim == 128 and c.speech_token_range == [10, 20] and c.channels == 8


def test_text_glue_matches_reference_fixture(gu, golden_dir):
    """tests/golden/text_glue.json holds outputs of the reference's own normalize_text /
    process_jsonl_item (tests/golden/make_golden_text.py), incl. the texts of its example jsonl files."""
    import json
    import os
    z = json.load(open(os.path.join(golden_dir, "text_glue.json")))
    assert len(z["normalize"]) > 20
    for src, want in z["normalize"]:
        assert gu.normalize_text(src) == want, src
    for item, want in z["items"]:
        assert gu.process_jsonl_item(dict(item)) == want, item
