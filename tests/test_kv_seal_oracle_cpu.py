"""The sealed KV page format as its specification states it (oracle/kv_seal_oracle.py; include/mtts.h: mtts_k_kv_seal):
seal -> unseal gives back every value of every lane that fits, V form and K form; the HIP sealer is held against the
same functions byte for byte in tests/test_kvpack_gpu.py."""
import numpy as np

from oracle import kv_seal_oracle as ks


def _bits(x):
    return (np.ascontiguousarray(x, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


def test_v_form_round_trip_and_fit_rate():
    rng = np.random.default_rng(0)
    fits = 0
    for i in range(6):
        x = rng.standard_normal((64, 128)).astype(np.float32) * np.float32(10.0 ** rng.integers(-3, 3))
        if i == 2:
            x[:, ::5] = 0.0
        page = _bits(x)
        sealed, fit = ks.seal(page)
        back, fit2 = ks.unseal(sealed)
        assert np.array_equal(fit, fit2)
        assert np.array_equal(back[fit], page[fit])
        fits += int(fit.sum())
    assert fits >= 0.99 * 6 * 64
    # more than 8 distinct high bytes in a lane: flagged, never approximated
    page = rng.integers(0, 1 << 16, (64, 128)).astype(np.uint16)
    sealed, fit = ks.seal(page)
    assert not fit.any() and (sealed[12, :, 12] == 1).all()


def test_k_form_undoes_any_per_dim_scale_and_flags_what_it_cannot_rescale():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((64, 128)).astype(np.float32) * (2.0 ** rng.integers(-12, 13, (1, 128))).astype(np.float32)
    page = _bits(x)
    assert not ks.seal(page)[1].any()                      # as it is: every token row spans far more than 16 binades
    sealed, fit = ks.seal(page, as_k=True)
    assert fit.mean() > 0.98
    back, _ = ks.unseal(sealed, as_k=True)
    assert np.array_equal(back[fit], page[fit])
    s = ks.k_shifts(page)
    assert np.array_equal(sealed[12, :, 8:10].copy().view(np.int8).reshape(128), s.astype(np.int8))
    page2 = page.copy()
    page2[3, 7] = 0x0001                                    # a denormal in a rescaled dim
    page2[4, 9] = 0xff80                                    # -inf
    page2[5, :] = page[5, :]
    _, fit2 = ks.seal(page2, as_k=True)
    assert not fit2[3] and not fit2[4]
    z = np.zeros((64, 128), dtype=np.uint16)                # an empty page seals to an all-zero dictionary, shifts 0
    sealed, fit = ks.seal(z, as_k=True)
    assert fit.all() and not sealed.any()


def test_v_form_undoes_per_token_scales():
    """A V page as its lanes see it (4 dims x 32 tokens per lane) with token norms spread over 16 binades and one very loud
    token: as it is, few lanes fit; divided per token by a power of two, nearly all do, and the values come back exactly."""
    rng = np.random.default_rng(2)
    x = rng.standard_normal((64, 128)).astype(np.float32) * (2.0 ** rng.integers(-8, 9, (64, 1))).astype(np.float32)
    x[5] *= 2.0 ** 20
    lanes = x.reshape(16, 2, 2, 32, 4).transpose(1, 3, 0, 4, 2).reshape(64, 128)   # x[token 4 it + 2 sub + h][dim 4 dl + c] -> lane 32 sub + dl, value 8 it + 2 c + h
    page = _bits(lanes)
    assert ks.seal(page)[1].mean() < 0.2
    sealed, fit = ks.seal(page, as_k=2)
    assert fit.mean() > 0.97
    back, _ = ks.unseal(sealed, as_k=2)
    assert np.array_equal(back[fit], page[fit])
    s = ks.v_shifts(page)
    assert s.min() == 0 and s[5] >= 15 and np.array_equal(sealed[12, :, 8], s.astype(np.uint8))
