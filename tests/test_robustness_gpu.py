"""Edge cases of the engine's host side: KV page pool recovery, eviction, chained finished-row resurrections.  -m gpu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from mtts import capi, synth  # noqa: E402
from oracle import asteroid_oracle as ao  # noqa: E402


def _text_prompt(cfg, rng, n, audio=0):
    raw = np.full((n, 8), 1024, dtype=np.int64)
    raw[:, 0] = rng.integers(0, 151643, n)
    if audio:
        raw[n - audio:, 0] = 151665 + rng.integers(0, 1024, audio)
        raw[n - audio:, 1:] = rng.integers(0, 1024, (audio, 7))
    return synth.shifting_inputs(raw, cfg["pad_token_id"])


def test_page_table_consistent_after_enomem_recovery():
    """A `begin` that runs out of KV pages half way through taking them (MTTS_ENOMEM) leaves page-table edits queued
    that never reached the device.  The next `begin` releases every slot and hands the same table indices out again
    with other page numbers: the stale edits must be gone, or one store would write two values to one entry and the
    device table could disagree with the host's (two dialogues aliasing one KV page).  Checked by reading the DEVICE
    table back and by the tokens of the run that follows (== the same run on a fresh engine)."""
    from mtts.engine import Engine
    cfg = synth.tiny()
    w = synth.synth_weights(cfg, 211, emb_row_sigma=0.6, speech_boost=6.0, eos_boost=1.0)
    eng = Engine(cfg, max_batch=4, max_seq_len=512, kv_pool_pages=9)
    eng.bind_state_dict(w)
    fresh = Engine(cfg, max_batch=4, max_seq_len=512, kv_pool_pages=9)
    fresh.bind_state_dict(w)
    big_ids, big_mask = synth.synth_prompts(cfg, 212, 4, 260, 0.3, False)      # 4 x 4 pages > 9: fails at the third row
    with pytest.raises(capi.MttsError) as ei:
        eng.begin(big_ids, big_mask, big_ids.shape[1] + 8)
    assert ei.value.code == capi.ENOMEM
    ids, mask = synth.synth_prompts(cfg, 213, 3, 100, 0.3, True)               # 3 x 2 pages fit
    ml = ids.shape[1] + 70                                                       # grows across a page boundary
    out = eng.generate(ids, mask, ml)
    eng.begin(ids, mask, ml)
    eng.step(40)
    eng.sync_state()
    host, n = eng.page_table(4)
    dev = eng.device_page_table(4)
    for b in range(3):
        assert n[b] >= 2
        assert np.array_equal(host[b, :n[b]], dev[b, :n[b]]), (b, host[b, :n[b]], dev[b, :n[b]])
    owned = [p for b in range(4) for p in host[b, :n[b]]]
    assert len(set(owned)) == len(owned)
    assert np.array_equal(out, fresh.generate(ids, mask, ml))
    # scheduler mode after a failed static begin
    with pytest.raises(capi.MttsError):
        eng.begin(big_ids, big_mask, big_ids.shape[1] + 8)
    eng.sched_open(3, 128)
    rng = np.random.default_rng(1)
    p = _text_prompt(cfg, rng, 90)
    eng.submit(1, p, p.shape[0] + 20, seed=5)
    host, n = eng.page_table(4)
    dev = eng.device_page_table(4)
    assert n[1] == 2 and np.array_equal(host[1, :2], dev[1, :2])
    eng.close()
    fresh.close()


def test_pool_runs_dry_eviction_path_equals_standalone():
    """4 slots, 13 pages: four short prompts (1 page each) are admitted together and each may grow to 5 pages, so the
    pool MUST run dry mid-flight: mtts_step -> MTTS_ENOMEM -> the youngest dialogue is evicted (mtts_slot_evict), its
    pages return, it is re-queued and re-run.  Evictions are asserted, every dialogue's tokens equal its batch-1 run,
    and every page is back in the pool afterwards."""
    from mtts.engine import Engine
    from mtts.scheduler import ContinuousBatcher
    cfg = synth.tiny()
    w = synth.synth_weights(cfg, 221, emb_row_sigma=0.6, speech_boost=6.0, eos_boost=1.0)     # nobody flushes early
    eng = Engine(cfg, max_batch=4, max_seq_len=384, kv_pool_pages=13)
    eng.bind_state_dict(w)
    solo = Engine(cfg, max_batch=1, max_seq_len=384)
    solo.bind_state_dict(w)
    rng = np.random.default_rng(9)
    prompts = [_text_prompt(cfg, rng, int(rng.integers(24, 50)), audio=6) for _ in range(7)]
    mnts = [int(rng.integers(230, 270)) for _ in range(7)]
    for layers, ds in ((None, None), ([dict(top_k=20, top_p=0.9, temperature=1.1)] * 8, [True] * 8)):
        cb = ContinuousBatcher(eng, slots=4, gen_cap=300, layers=layers, do_samples=ds, steps_per_poll=8)
        seeds = list(range(500, 507))
        got = cb.run(prompts, mnts, seeds=seeds)
        assert cb.evictions > 0, "the pool never ran dry: the eviction path did not execute"
        for i, p in enumerate(prompts):
            alone = solo.generate(p[None], np.ones((1, p.shape[0])), p.shape[0] + mnts[i], layers=layers, do_samples=ds,
                                  seed=seeds[i])[0]
            assert alone.shape[0] - (p.shape[0] - 7) >= 200                  # long enough to need 4+ pages
            assert got[i].shape == alone.shape and np.array_equal(got[i], alone), i
        total, free, _ = eng.kv_pool_state()
        assert free == total == 13
        print("eviction run: evictions", cb.evictions, "engine steps", cb.engine_steps)
    eng.close()
    solo.close()


def test_chained_resurrections_run_past_max_length_plus_14():
    """The reference re-tests EVERY cut-off row on every step while any flush is still running
    (modeling_asteroid.py:140-141,168), so resurrections chain: row 0's flush starts 2 steps before max_length and keeps
    the batch alive until max_length + 5; inside it (step M+3) row 1 -- cut off by max_length -- picks a non-speech
    token and is resurrected for a flush that lasts until M+10; inside THAT (step M+8) row 2 is resurrected and
    flushes until M+15: one step more than a single resurrection can reach (14).  The draws are scripted
    (forced_as_draw="all": the forced row is every row's raw pick, finished rows included) and the oracle -- whose
    loop is the reference's line by line, pinned by ar_flush_past_max.npz -- gives the expected rows."""
    from mtts.engine import Engine
    cfg = synth.tiny()
    w = synth.synth_weights(cfg, 231, emb_row_sigma=0.6, speech_boost=6.0, eos_boost=1.0)
    ids, mask = synth.synth_prompts(cfg, 232, 3, 24, 0.0, False)
    T = ids.shape[1]
    base = T - 7
    M = 12                                                   # max_length - base
    max_length = base + M
    G = M + 26
    rng = np.random.default_rng(3)
    forced = np.zeros((3, base + G, 8), dtype=np.int64)
    forced[:, :base] = ids[:, :base]
    forced[:, base:, 0] = 151665 + rng.integers(0, 1024, (3, G))          # speech picks ...
    forced[:, base:, 1:] = rng.integers(0, 1024, (3, G, 7))
    nonspeech = 77
    forced[0, base + M - 2, 0] = nonspeech                   # ... except: row 0 starts its flush at step M-2,
    forced[1, base + M + 3, 0] = nonspeech                   # row 1 is resurrected at M+3 (row 0 flushes until M+5),
    forced[2, base + M + 8, 0] = nonspeech                   # row 2 at M+8 (row 1 flushes until M+10)
    orc = ao.AsteroidOracle(cfg, w, "bf16")
    want, odec, _ = orc.generate(ids, mask, max_length, forced=forced, forced_as_draw="all")
    steps = want.shape[1] - base
    assert steps == M + 15, steps                            # the chain did run past max_length + 14
    eos = cfg["eos_token_id"]
    assert want[2, base + M + 14, 0] == eos and want[2, base + M + 9, 0] == eos and want[2, base + M + 7, 0] == eos
    assert (want[2, base + M + 14, 1:] == 1024).sum() == 6  # last flush row of row 2: only channel 7 still carries a code
    eng = Engine(cfg, max_batch=4, max_seq_len=256)
    eng.bind_state_dict(w)
    out, dec = eng.generate(ids, mask, max_length, forced=forced, forced_as_draw="all")
    assert out.shape == want.shape, (out.shape, want.shape)
    assert np.array_equal(out, want)
    eng.close()
