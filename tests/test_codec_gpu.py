"""HIP codec decoder (C ABI) vs the numpy oracle and the reference-generated fixtures.  -m gpu."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from mtts import capi, synth_codec  # noqa: E402
from oracle import codec_oracle as co  # noqa: E402

RMS_TOL = 1e-4


def test_gemm_f32_kernel():
    from mtts import codec as mc
    lib = capi.lib()
    rng = np.random.default_rng(0)
    for (M, N, K, act) in [(375, 3072, 512, 0), (3000, 512, 4096, 0), (130, 962, 512, 1), (64, 64, 16, 0)]:
        a = rng.standard_normal((M, K)).astype(np.float32)
        w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32)
        at, wt, bt = (torch.from_numpy(x).cuda() for x in (a, w, b))
        c = torch.zeros(M, N, device="cuda")
        mc._check(lib.mtts_k_gemm_f32(at.data_ptr(), wt.data_ptr(), bt.data_ptr(), c.data_ptr(), M, N, K, act, None))
        torch.cuda.synchronize()
        ref = a.astype(np.float64) @ w.T.astype(np.float64) + b
        if act:
            from scipy.special import erf
            ref = 0.5 * ref * (1 + erf(ref / np.sqrt(2)))
        np.testing.assert_allclose(c.cpu().numpy(), ref, rtol=2e-5, atol=2e-5)
        # the decode direction's kernel: every product as 3 bf16 MFMAs (hi*hi + hi*lo + lo*hi), fp32 accumulate.
        # Dropped terms are < 2^-16 of a product: error bound relative to sum |a||w| (not to the result's size)
        c3 = torch.zeros(M, N, device="cuda")
        mc._check(lib.mtts_k_gemm_f32(at.data_ptr(), wt.data_ptr(), bt.data_ptr(), c3.data_ptr(), M, N, K, act | 0x100, None))
        torch.cuda.synchronize()
        bound = 3e-5 * (np.abs(a).astype(np.float64) @ np.abs(w).T.astype(np.float64)) + 1e-6
        assert (np.abs(c3.cpu().numpy() - ref) <= bound).all()
        assert np.abs(c3.cpu().numpy() - ref).max() < 2e-4


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = json.loads(str(z["cfg"]))
    w = synth_codec.synth_weights(cfg, int(z["seed"]))
    codes = synth_codec.synth_codes(cfg, int(z["seed"]) + 1, list(z["lengths"]))
    return z, cfg, w, codes


@pytest.mark.parametrize("name,gemm", [("codec_T40", None), ("codec_ragged_1win", None), ("codec_T600", None),
                                       ("codec_full_T24", None), ("codec_ragged_1win", "f32"), ("codec_full_T24", "f32")])
def test_codec_matches_reference_fixture(golden_dir, monkeypatch, name, gemm):
    """Default: bf16x3 GEMMs + fused attention; gemm="f32": the exact-f32 kernels (MTTS_CODEC_GEMM=f32)."""
    from mtts.codec import CodecEngine
    if gemm:
        monkeypatch.setenv("MTTS_CODEC_GEMM", gemm)
    z, cfg, w, codes = _load(golden_dir, name)
    eng = CodecEngine(cfg)
    eng.bind_state_dict(w)
    wavs = [x.cpu().numpy() for x in eng.decode([torch.from_numpy(c) for c in codes])]
    eng.close()
    stride = int(z["stride"])
    for i, wv in enumerate(wavs):
        assert wv.shape[0] == int(z[f"wav{i}_len"])
        err = wv[::stride].astype(np.float64) - z[f"wav{i}_sub"].astype(np.float64)
        rms_err = float(np.sqrt(np.mean(err ** 2)))
        assert rms_err <= RMS_TOL, (name, i, rms_err)          # north_star: waveform RMS within 1e-4
        assert float(np.abs(err).max()) <= 1e-3
        sec = 24000
        rms = np.array([np.sqrt(np.mean(wv[s:s + sec].astype(np.float64) ** 2)) for s in range(0, wv.shape[0], sec)])
        np.testing.assert_allclose(rms, z[f"wav{i}_rms"], atol=RMS_TOL)


def test_codec_matches_oracle_full_waveform_and_edges():
    from mtts.codec import CodecEngine
    cfg = synth_codec.reduced(dec_layers=1, voc_layers=2)
    w = synth_codec.synth_weights(cfg, 21)
    eng = CodecEngine(cfg)
    eng.bind_state_dict(w)
    orc = co.CodecOracle(cfg, w)
    for lengths in ([1], [7, 3], [251]):
        codes = synth_codec.synth_codes(cfg, 22, lengths)
        got = [x.cpu().numpy() for x in eng.decode([torch.from_numpy(c) for c in codes])]
        want = orc.decode(codes)
        for g, wv, n in zip(got, want, lengths):
            assert g.shape[0] == n * 1920
            e = g.astype(np.float64) - wv.astype(np.float64)
            assert np.sqrt(np.mean(e ** 2)) <= RMS_TOL
    # out-of-range code index is rejected loudly
    bad = synth_codec.synth_codes(cfg, 23, [5])
    bad[0][3, 2] = 4096
    with pytest.raises(capi.MttsError):
        eng.decode([torch.from_numpy(bad[0])])
    eng.close()


@pytest.mark.parametrize("name", ["codec_enc_3s", "codec_enc_ragged", "codec_enc_35s"])
def test_codec_encode_matches_reference_fixture(golden_dir, name):
    """Exact code ids vs the reference XY_Tokenizer.encode (CPU) on the same synthetic audio."""
    from mtts.codec import CodecEngine
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = json.loads(str(z["cfg"]))
    w = synth_codec.synth_weights(cfg, int(z["seed"]), encoder=True)
    wavs = synth_codec.synth_wavs(int(z["seed"]) + 1, list(z["lengths"]))
    eng = CodecEngine(cfg)
    eng.bind_state_dict(w)
    got = [g.cpu().numpy() for g in eng.encode([torch.from_numpy(x) for x in wavs])]
    eng.close()
    for i, g in enumerate(got):
        want = z[f"codes{i}"].astype(np.int64)
        assert g.shape == want.shape
        mism = float((g != want).mean())
        assert mism == 0.0, (name, i, mism)


def test_codec_encode_decode_roundtrip_shapes():
    """encode -> decode round trip keeps the length bookkeeping (1280 in / 1920 out per code)."""
    from mtts.codec import CodecEngine
    cfg = synth_codec.reduced(dec_layers=1, voc_layers=1, enc_layers=1)
    w = synth_codec.synth_weights(cfg, 31, encoder=True)
    eng = CodecEngine(cfg)
    eng.bind_state_dict(w)
    wavs = synth_codec.synth_wavs(32, [16000 * 2 + 123, 1280 * 5])
    codes = eng.encode([torch.from_numpy(x) for x in wavs])
    assert [tuple(c.shape) for c in codes] == [(8, (16000 * 2 + 123) // 1280), (8, 5)]
    out = eng.decode(codes)
    assert [o.shape[0] for o in out] == [c.shape[1] * 1920 for c in codes]
    eng.close()


def test_codec_decode_each_equals_per_sample_calls():
    """CodecEngine.decode_each runs the 30 s windows of ALL sequences in shared calls (what process_batch uses) while
    keeping the reference pipeline's semantics of one `spt.decode([codes])` per sample (generation_utils.py:434-450):
    only windows of equal length share a call.  Seven sequences of 13..760 codes (1..3 windows, ragged last windows,
    two of equal length): identical waveforms, for two call sizes."""
    from mtts.codec import CodecEngine
    cfg = synth_codec.reduced(dec_layers=2, voc_layers=2)
    w = synth_codec.synth_weights(cfg, 41)
    eng = CodecEngine(cfg)
    eng.bind_state_dict(w)
    rng = np.random.default_rng(42)
    codes = [torch.from_numpy(rng.integers(0, 1024, (8, n))) for n in (40, 375, 760, 13, 410, 760, 250)]
    solo = [eng.decode([c])[0].cpu().numpy() for c in codes]
    for wpc in (32, 3):
        both = [x.cpu().numpy() for x in eng.decode_each(codes, windows_per_call=wpc)]
        for a, b in zip(solo, both):
            assert a.shape == b.shape
            rms = float(np.sqrt(np.mean((a.astype(np.float64) - b) ** 2)))
            assert rms <= 1e-6, (wpc, rms)
    eng.close()


def test_codec_gemm_tile_variants_are_bit_identical(monkeypatch):
    """gemm_b3t_kernel exists in ten tile shapes (32x32 .. 128x128 outputs per wave, one or two waves per SIMD) and
    the host picks one per GEMM shape, so the choice changes with the number of windows in a call.  A tile shape only
    decides which wave owns an output element: every variant must give the same bits.  Ragged lengths (rows not a
    multiple of 32 or of any tile) through the whole decoder, each variant forced in turn (MTTS_CODEC_TILE)."""
    from mtts.codec import CodecEngine
    cfg = synth_codec.reduced(dec_layers=1, voc_layers=2)
    w = synth_codec.synth_weights(cfg, 77)
    rng = np.random.default_rng(78)
    codes = [torch.from_numpy(rng.integers(0, 1024, (8, n))) for n in (13, 131, 375)]
    outs = {}
    for code in ("0", "2222", "2312", "4221", "3311", "3411", "4311", "4411", "2122", "1222", "1122"):
        monkeypatch.setenv("MTTS_CODEC_TILE", code)
        eng = CodecEngine(cfg)
        eng.bind_state_dict(w)
        outs[code] = [x.cpu().numpy() for x in eng.decode_each(codes)]
        eng.close()
    for code, got in outs.items():
        for a, b in zip(outs["0"], got):
            assert a.shape == b.shape and np.array_equal(a, b), code
    assert all(np.isfinite(x).all() and float(np.abs(x).max()) > 0 for x in outs["0"])


def test_gemm_planes_bench_hook_runs_every_variant():
    """tools/gemm_planes_bench.py / gemm_planes_pmc.sh rest on mtts_k_gemm_planes_bench: every tile code and every epilogue
    combination the decoder uses launches and reports a time; an unknown combination is refused."""
    import ctypes as C
    lib = capi.lib()
    us, used = C.c_float(), C.c_int32()
    for code in (0, 2222, 2312, 4221, 3311, 3411, 4311, 4411, 2122, 1222, 1122):
        for flags in (0, 4, 6, 9):
            assert lib.mtts_k_gemm_planes_bench(1000, 768, 256, flags, code, 2, C.byref(us), C.byref(used)) == 0
            assert us.value > 0 and (code == 0 or used.value == code)
    assert lib.mtts_k_gemm_planes_bench(1000, 768, 256, 3, 0, 2, C.byref(us), C.byref(used)) != 0
