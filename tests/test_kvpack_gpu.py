"""Sealed KV pages (csrc/attn.hip: seal_lane / pk_unit; include/mtts.h: mtts_k_kv_seal).  -m gpu.

A complete KV page is kept a second time in a 13-bit form the decode attention reads instead of the bf16 page.  The form
is LOSSLESS, so everything here is bit-for-bit: the format against its documented layout (decoded with numpy), the
attention launches on sealed pages against the same launches on bf16 pages, the engine with the switch on against off."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from mtts import capi, synth  # noqa: E402
from oracle import asteroid_oracle as ao  # noqa: E402
from oracle import kv_seal_oracle as ks  # noqa: E402


def _bf16_bits(x):
    return (np.ascontiguousarray(x, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


def _decode_sealed(sealed):
    """numpy restatement of the documented layout: sealed uint8 [pages, 13, 64, 16] -> (uint16 [pages, 64, 128] in the
    lane's value order, flags [pages, 64])."""
    P = sealed.shape[0]
    low = sealed[:, 0:8].transpose(0, 2, 1, 3).reshape(P, 64, 128).astype(np.uint16)
    nib = sealed[:, 8:12].transpose(0, 2, 1, 3).reshape(P, 64, 16, 4)            # [.., j, k]
    code = np.concatenate([nib & 0xf, nib >> 4], axis=-1).reshape(P, 64, 128)     # value 8j + k / 8j + 4 + k
    last = sealed[:, 12]                                                          # [P, 64, 16]
    dic = last[..., :8].astype(np.uint16)
    flag = last[..., 12:16].copy().view(np.uint32)[..., 0]
    hi = np.take_along_axis(dic, (code & 7).astype(np.int64), axis=-1)
    return ((code.astype(np.uint16) >> 3) << 15) | (hi << 8) | low, flag


def _k_shifts(sealed):
    """The per-dim shifts a K page carries: lane l keeps s[2l], s[2l+1] as int8 in its spare -> int [pages, 128]."""
    return sealed[:, 12, :, 8:10].copy().view(np.int8).reshape(sealed.shape[0], 128).astype(np.int32)


def _seal(pages_u16, as_k=0):
    """uint16 [pages, 64 lanes, 128 values] (lane order) -> sealed uint8 [pages, 13, 64, 16] through the HIP kernel."""
    P = pages_u16.shape[0]
    raw = pages_u16.reshape(P, 64, 16, 8).transpose(0, 2, 1, 3)                    # [page][unit][lane][8 halves]
    t = torch.from_numpy(np.ascontiguousarray(raw).view(np.int16)).cuda()
    out = torch.zeros(P, 13, 64, 16, dtype=torch.uint8, device="cuda")
    capi.check(capi.lib().mtts_k_kv_seal(t.data_ptr(), P, out.data_ptr(), int(as_k), None))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_sealed_page_format_is_lossless_and_flags_what_does_not_fit():
    rng = np.random.default_rng(5)
    P = 12
    x = rng.standard_normal((P, 64, 128)).astype(np.float32)
    x[1] *= 1e-3
    x[2] *= 300.0
    x[3, :, ::7] = 0.0                                      # exact zeros next to normal values (RoPE cancellations)
    x[4, :, 5] = -0.0
    x[5] = np.where(rng.random((64, 128)) < 0.5, x[5], x[5] * 2.0 ** -9)   # two clusters of exponents
    v = _bf16_bits(x)
    # lanes built by hand: exactly 8 distinct high bytes (fits), 9 (does not), denormals, inf / nan patterns
    e8 = np.array([0x3f, 0x3e, 0x3d, 0x3c, 0x3b, 0x20, 0x10, 0x00], dtype=np.uint16)
    v[6, 0] = (e8[rng.integers(0, 8, 128)] << 8) | rng.integers(0, 256, 128) | (rng.integers(0, 2, 128) << 15)
    v[6, 0, :8] = (e8 << 8) | 0x55
    v[6, 1] = v[6, 0]
    v[6, 1, 9] = 0x2a11                                     # a ninth high byte
    v[6, 2] = rng.integers(0, 0x80, 128)                    # denormals: high byte 0
    v[6, 3] = np.where(rng.random(128) < 0.5, 0x7f80, 0xffc1).astype(np.uint16)   # +inf and a negative NaN: high byte 0x7f
    v[7] = rng.integers(0, 1 << 16, (64, 128))              # random bits: ~100 distinct high bytes per lane
    got, flag = _decode_sealed(_seal(v))
    assert flag[7].all()                                    # nothing of the random-bits page fits
    assert flag[6, 1] == 1 and flag[6, 0] == 0 and flag[6, 2] == 0 and flag[6, 3] == 0
    fits = flag == 0
    assert fits[:5].mean() > 0.99                           # Gaussian lanes fit (a lane needs 9+ distinct exponent pairs not to)
    assert np.array_equal(got[fits], v[fits])               # bit for bit, -0.0 / inf / nan / denormals included
    # the dictionary is sorted ascending and holds exactly the distinct high bytes
    s = _seal(v)
    for pg, ln in [(0, 0), (6, 0), (3, 17)]:
        want = np.unique((v[pg, ln] >> 8) & 0x7f)
        assert list(s[pg, 12, ln, :len(want)]) == list(want)


def test_sealed_k_pages_rescale_each_dim_by_a_power_of_two_exactly():
    """K pages (lane = token, value = dim): dims whose scales differ by up to 2^24 (a k_norm weight would do that) still
    seal, because each dim is divided by a power of two kept in the page; stored value x 2^s[d] is the original, bit for bit."""
    rng = np.random.default_rng(9)
    P = 8
    x = rng.standard_normal((P, 64, 128)).astype(np.float32)
    x *= (2.0 ** rng.integers(-12, 13, (P, 1, 128))).astype(np.float32)
    x[1, :, 5] = 0.0                                        # an all-zero dim: shift 0
    x[2, ::3, 7] = 0.0                                      # zeros inside a scaled dim stay zeros
    v = _bf16_bits(x)
    v[3, 10, 20] = 0x0003                                   # a denormal in a rescaled dim: that lane cannot be exact
    v[4, 11, 21] = 0x7f80                                   # +inf likewise
    v[5, :, 30] = _bf16_bits(np.full(64, 2.0 ** -100, np.float32))
    v[5, 7, 30] = _bf16_bits(np.array([2.0 ** 100], np.float32))[0]    # one dim spanning 200 binades: the outlying token does not fit
    sealed = _seal(v, as_k=1)
    got, flag = _decode_sealed(sealed)
    sh = _k_shifts(sealed)                                  # [P, 128]
    e = ((v >> 7) & 0xff).astype(np.int64)
    cnt, tot = (e > 0).sum(axis=1), e.sum(axis=1)
    assert np.array_equal(sh, np.where(cnt > 0, np.clip((tot + cnt // 2) // np.maximum(cnt, 1) - 125, -127, 127), 0))
    assert flag[3, 10] == 1 and flag[4, 11] == 1 and flag[5].sum() == 1 and flag[5, 7] == 1
    fits = flag == 0
    assert fits[[0, 1, 2, 6, 7]].mean() > 0.99              # the 24-binade spread between dims is gone
    nz = (got & 0x7fff) != 0
    back = np.where(nz, (got.astype(np.int32) + (sh[:, None, :] << 7)) & 0xffff, got).astype(np.uint16)
    assert np.array_equal(back[fits], v[fits])
    assert (_seal(v, as_k=0)[:, 12, :, 12:16].view(np.uint32)[..., 0] != 0).mean() > 0.9    # without the rescale nothing fits


@pytest.mark.parametrize("as_k", [0, 1, 2])
def test_hip_sealer_equals_the_format_oracle_byte_for_byte(as_k):
    """csrc/attn.hip: seal_lane / seal_lane_k against oracle/kv_seal_oracle.py (the format's specification made
    executable): every byte of every lane that fits, and the dictionary / shifts / flag unit of every lane."""
    rng = np.random.default_rng(21 + as_k)
    P = 10
    x = rng.standard_normal((P, 64, 128)).astype(np.float32)
    x[1] *= (2.0 ** rng.integers(-12, 13, (1, 128))).astype(np.float32)
    x[2] *= (2.0 ** rng.integers(-3, 4, (64, 128))).astype(np.float32)
    x[9] *= (2.0 ** rng.integers(-10, 11, (64, 1))).astype(np.float32)      # (as a V page this is NOT one scale per token: lanes mix tokens)
    x[3, :, ::9] = 0.0
    x[4] *= 1e-30
    x[5] *= 1e30
    v = _bf16_bits(x)
    v[6] = rng.integers(0, 1 << 16, (64, 128))
    v[7, 5, 17] = 0x0002
    v[7, 9, 3] = 0x7fc0
    v[8] = 0
    got = _seal(v, as_k=as_k)                                    # [P, 13, 64, 16]
    nfit = 0
    for pg in range(P):
        want, fit = ks.seal(v[pg], as_k=as_k)
        assert np.array_equal(got[pg, 12], want[12]), pg         # dictionary, shifts, flag: every lane
        assert np.array_equal(got[pg][:, fit, :], want[:, fit, :]), pg
        back, fit2 = ks.unseal(got[pg], as_k=as_k)
        assert np.array_equal(fit, fit2) and np.array_equal(back[fit], v[pg][fit])
        nfit += int(fit.sum())
    assert nfit > 0.5 * P * 64


@pytest.mark.parametrize("nq,nkv", [(4, 2), (8, 2), (4, 4)])
def test_decode_attention_on_sealed_pages_equals_bf16_pages_bit_for_bit(monkeypatch, nq, nkv):
    """The three decode launches over a shuffled page table, ragged lengths, with rows that seal and rows that do not
    (wide exponent ranges in K and in V): sealed pages on == off, every bit."""
    lib = capi.lib()
    rng = np.random.default_rng(100 + nq)
    R, Lmax = 6, 1100
    lens = np.array([1100, 1, 64, 65, 777, 1024], dtype=np.int32)
    q = ao.round_bf16(rng.standard_normal((R, nq, 128)).astype(np.float32))
    K = rng.standard_normal((R, Lmax, nkv, 128)).astype(np.float32)
    V = rng.standard_normal((R, Lmax, nkv, 128)).astype(np.float32)
    K[0, 100:140] *= (2.0 ** rng.integers(-40, 1, (40, nkv, 128))).astype(np.float32)      # tokens whose K row cannot seal
    V[4, 200:300, :, 8:12] *= (2.0 ** rng.integers(-40, 1, (100, nkv, 4))).astype(np.float32)   # a V lane that cannot
    K[1:, :, :, 3] = 0.0
    K[5] *= (2.0 ** rng.integers(-12, 13, (1, nkv, 128))).astype(np.float32)               # dims on very different scales: seals (rescaled)
    V[3] *= (2.0 ** rng.integers(-10, 11, (Lmax, 1, 1))).astype(np.float32)                 # tokens on very different scales: V seals (rescaled per token)
    V[5, 500:520] *= np.float32(2.0 ** -110)                                                 # tokens so quiet that the rescaled loud ones cannot stay normal
    q[2] *= 2.0 ** 60                                                                         # a q the rescale cannot take everywhere: bf16 pages
    K, V = ao.round_bf16(K), ao.round_bf16(V)
    pages = (Lmax + 63) // 64
    table = rng.permutation(R * pages).astype(np.int32).reshape(R, pages)
    bf = lambda a: torch.from_numpy(_bf16_bits(a).view(np.int16)).cuda().view(torch.bfloat16)
    qt, kt, vt = bf(q), bf(K), bf(V)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MTTS_KV_PACK", mode)
        out = torch.zeros(R, nq * 128, dtype=torch.bfloat16, device="cuda")
        capi.check(lib.mtts_k_paged_attn_decode(qt.data_ptr(), kt.data_ptr(), vt.data_ptr(), lens.ctypes.data, table.ctypes.data,
                                                R, Lmax, nq, nkv, out.data_ptr(), None))
        torch.cuda.synchronize()
        outs[mode] = out.view(torch.int16).cpu().numpy()
    assert np.array_equal(outs["0"], outs["1"])
    assert np.abs(outs["1"]).max() > 0


def test_engine_with_sealed_pages_equals_engine_without_and_pages_do_seal(monkeypatch):
    """A 200-token prompt (3 complete pages from the prefill) + 150 decode steps (2 more sealed on the way), B = 3 ragged:
    same tokens, same logits bits with MTTS_KV_PACK on and off; the counters show the pages exist and seal."""
    from mtts.engine import Engine
    cfg = synth.tiny()
    w = synth.synth_weights(cfg, 77, emb_row_sigma=0.6, speech_boost=5.0)
    rng = np.random.default_rng(3)
    B, T = 3, 200
    ids = np.full((B, T, 8), 1024, dtype=np.int64)
    ids[:, :, 0] = rng.integers(0, cfg["vocab_size"] - 10, (B, T))
    mask = np.ones((B, T), dtype=np.int64)
    ids[1, :37] = [cfg.get("pad_token_id", 0)] + [1024] * 7
    mask[1, :37] = 0
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MTTS_KV_PACK", mode)
        eng = Engine(cfg, max_batch=4, max_seq_len=512)
        eng.bind_state_dict(w)
        out = eng.generate(ids, mask, T + 150, do_samples=[False] * 8)
        l0, l17 = eng.read_logits()
        st = eng.kv_pack_stats() if mode == "1" else None
        res[mode] = (out, np.asarray(l0).copy(), np.asarray(l17).copy(), st)
        if mode == "0":
            with pytest.raises(capi.MttsError):
                eng.kv_pack_stats()
        eng.close()
    assert np.array_equal(res["0"][0], res["1"][0])
    assert np.array_equal(res["0"][1], res["1"][1]) and np.array_equal(res["0"][2], res["1"][2])
    st = res["1"][3]
    os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r03_kv_pack_stats_tiny.json"), "w") as f:
        json.dump(st, f)
    assert st["k_pages"] > 0 and st["k_pages"] == st["v_pages"]
    assert st["k_unsealed"] <= 0.05 * st["k_pages"] and st["v_unsealed"] <= 0.05 * st["v_pages"], st


def test_k_seals_under_any_k_norm_spread_and_the_read_policy_drops_v_when_v_does_not(monkeypatch):
    """k_norm weights spread over 24 binades (the same for the two dims RoPE rotates into each other: a ratio between
    THOSE makes a dim sweep through all the binades in between as the angle turns, which no per-dim scale can undo):
    a K row then holds far more than 8 distinct exponent pairs, but the sealer
    rescales each dim by a power of two (and the score kernel q by the same), so K still seals.  v_proj rows scaled the
    same way: a V lane (4 neighbouring dims x 32 tokens) cannot seal; after the first host sync that has seen 16 pages the
    engine stops reading the sealed V pages of those layers.  Results stay those of the bf16 pages throughout."""
    from mtts.engine import Engine
    cfg = synth.tiny()
    w = synth.synth_weights(cfg, 78, emb_row_sigma=0.6, speech_boost=5.0)
    rng = np.random.default_rng(4)
    kn = [k for k in w if k.endswith("k_norm.weight")]
    vp = [k for k in w if k.endswith("v_proj.weight")]
    assert len(kn) == cfg["num_hidden_layers"] == len(vp)
    for k in kn:
        sc = np.tile(2.0 ** rng.integers(-12, 13, 64), 2).astype(np.float32)   # RoPE partners (d, d + 64) share a scale; powers of two: exact bf16
        w[k] = (w[k] * sc).astype(np.float32)
    for k in vp:
        w[k] = (w[k] * (2.0 ** (-8.0 * (np.arange(w[k].shape[0]) % 4)))[:, None].astype(np.float32)).astype(np.float32)   # a lane's 4 dims: 1, 2^-8, 2^-16, 2^-24
    B, T = 2, 700
    ids = np.full((B, T, 8), 1024, dtype=np.int64)
    ids[:, :, 0] = rng.integers(0, cfg["vocab_size"] - 10, (B, T))
    mask = np.ones((B, T), dtype=np.int64)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MTTS_KV_PACK", mode)
        eng = Engine(cfg, max_batch=2, max_seq_len=1024)
        eng.bind_state_dict(w)
        out = eng.generate(ids, mask, T + 40, do_samples=[False] * 8)
        l0, l17 = eng.read_logits()
        res[mode] = (out, np.asarray(l0).copy(), np.asarray(l17).copy(), eng.kv_pack_stats() if mode == "1" else None)
        eng.close()
    assert np.array_equal(res["0"][0], res["1"][0])
    assert np.array_equal(res["0"][1], res["1"][1]) and np.array_equal(res["0"][2], res["1"][2])
    st = res["1"][3]
    L = cfg["num_hidden_layers"]
    assert st["k_pages"] >= 2 * 10 * cfg["num_key_value_heads"] * L
    assert st["k_unsealed"] <= 0.1 * st["k_pages"] and st["k_layers_on"] == L, st
    assert st["v_unsealed"] >= 0.9 * st["v_pages"] and st["v_layers_on"] == 0, st


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_ragged_prompts_around_page_boundaries_sampled_runs_equal_with_and_without_sealed_pages(monkeypatch, seed):
    """Prompts of 56..136 real tokens (pages complete during the prefill, at the first decode steps, in the middle of
    the run), left-padded to one batch, SAMPLED with top-k / top-p / temperature / repetition penalty: every token of
    every row identical with sealed pages on and off, through page completions at different steps in different rows."""
    from mtts.engine import Engine
    cfg = synth.tiny()
    w = synth.synth_weights(cfg, 300 + seed, emb_row_sigma=0.6, speech_boost=5.0)
    rng = np.random.default_rng(seed)
    lens = [56, 57, 63, 64, 65, 120, 121, 136]                 # real tokens incl. the 7 delay slots
    B, T = len(lens), max(lens)
    ids = np.full((B, T, 8), 1024, dtype=np.int64)
    ids[:, :, 0] = rng.integers(0, cfg["vocab_size"] - 10, (B, T))
    mask = np.zeros((B, T), dtype=np.int64)
    for b, n in enumerate(lens):
        mask[b, T - n:] = 1
        ids[b, :T - n, 0] = cfg.get("pad_token_id", 0)
    layers = [{"top_k": 20, "top_p": 0.9, "temperature": 1.1, "repetition_penalty": 1.05}] * 8
    outs = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("MTTS_KV_PACK", mode)
        eng = Engine(cfg, max_batch=B, max_seq_len=512)
        eng.bind_state_dict(w)
        outs[mode] = eng.generate(ids, mask, T + 150, layers=layers, do_samples=[True] * 8, seed=77 + seed)
        if mode == "2":
            st = eng.kv_pack_stats()
        eng.close()
    assert outs["0"].shape == outs["2"].shape and np.array_equal(outs["0"], outs["2"])
    assert outs["0"].shape[1] > T + 60                          # the rows did run (no early EOS for everyone)


def test_continuous_batching_reuses_physical_pages_whose_sealed_copies_are_stale(monkeypatch):
    """7 dialogues with 70-200-token prompts and 60-130 new tokens through 2 slots and a pool of 14 pages: slots are
    refilled while the other is mid-flight and physical pages go from one dialogue to the next with the previous owner's
    sealed copy still in the second pool.  A page is only read sealed once ITS CURRENT owner has completed (and so
    re-sealed) it: every dialogue's tokens equal its batch-1 run on bf16 pages (sampled, per-dialogue Philox keys)."""
    from mtts.engine import Engine
    from mtts.scheduler import ContinuousBatcher
    cfg = synth.tiny()
    w = synth.synth_weights(cfg, 193, emb_row_sigma=0.6, speech_boost=5.0)
    rng = np.random.default_rng(17)
    prompts, mnts = [], []
    for i in range(7):
        n = int(rng.integers(70, 200))
        raw = np.full((n, 8), 1024, dtype=np.int64)
        raw[:, 0] = rng.integers(0, 151643, n)
        prompts.append(synth.shifting_inputs(raw, cfg["pad_token_id"]))
        mnts.append(int(rng.integers(60, 130)))
    layers, ds = [dict(top_k=20, top_p=0.9, temperature=1.1)] * 8, [True] * 8
    monkeypatch.setenv("MTTS_KV_PACK", "2")
    eng = Engine(cfg, max_batch=2, max_seq_len=512, kv_pool_pages=14)
    eng.bind_state_dict(w)
    cb = ContinuousBatcher(eng, slots=2, gen_cap=160, layers=layers, do_samples=ds, steps_per_poll=4)
    got = cb.run(prompts, mnts, base_seed=90)
    eng.close()
    monkeypatch.setenv("MTTS_KV_PACK", "0")
    solo = Engine(cfg, max_batch=1, max_seq_len=512)
    solo.bind_state_dict(w)
    for i, p in enumerate(prompts):
        alone = solo.generate(p[None], np.ones((1, p.shape[0])), p.shape[0] + mnts[i], layers=layers, do_samples=ds, seed=90 + i)[0]
        assert got[i].shape == alone.shape and np.array_equal(got[i], alone), i
    solo.close()
