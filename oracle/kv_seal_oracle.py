"""CPU restatement of the sealed KV page format (numpy).

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (`moss-ttsd_amd/`) may import this module; `tests/` use it
as the checker of `csrc/attn.hip: seal_lane / seal_lane_k / pk_unit`.

The format has no counterpart in the reference (the reference's KV cache is plain bf16, transformers `DynamicCache`
as used by modeling_asteroid.py:337-426): it is this engine's own LOSSLESS second copy of a complete 64-token page, so
the oracle here is the format's specification (include/mtts.h: mtts_k_kv_seal) made executable: `seal()` must produce
the bytes the HIP sealer produces, `unseal()` must give back the page, bit for bit.

Page as the cache holds it: [16 units][64 lanes][16 B]; a lane's 128 values in order = its 16 units x 8 bf16.
Sealed page: [13 units][64 lanes][16 B]:
  units 0-7   the low bytes of the lane's 128 values, in order
  units 8-11  one code nibble per value: byte 4j+k holds value 8j+k (low nibble) and value 8j+4+k (high nibble);
              code = sign << 3 | index into the lane's dictionary
  unit 12     8 dictionary bytes ((bf16 >> 8) & 0x7f of the values present, ascending, zero-padded), 4 spare bytes,
              a 32-bit flag (1: the lane did not fit -- more than 8 distinct entries, or a K value that cannot be rescaled --
              and units 0-11 of that lane are undefined)
K pages (lane = token, value i = dim i): before the above, dim d is divided by 2^s[d] with
  s[d] = clip(round-half-up(mean of the NON-ZERO exponent fields of dim d over the 64 tokens) - 125, -127, 127), 0 if none;
  lane l keeps s[2l], s[2l+1] (int8) in the first two spare bytes.  A value with exponent field 0 stays as it is (and
  makes its lane unfit if it is a denormal and s != 0); exponent field 255 makes its lane unfit if s != 0; otherwise the
  new exponent field e - s must lie in 1..254 or the lane is unfit.
V pages (lane = 32 sub + dl; its value 8 it + 2 c + h is token 4 it + 2 sub + h, dim 4 dl + c): before the above, token t is
  divided by 2^s[t] with s[t] = min((m[t] - min over tokens with m > 0 of m) rounded down to even, 126), m[t] = round-half-up(mean of the non-zero
  exponent fields of token t's 128 values) (s = 0 for an all-zero token); lane t keeps s[t] in the first spare byte; the same
  rules for exponent fields 0 and 255, and e - s >= 1.  The reader multiplies token t's probability by 2^s[t]."""
import numpy as np


def k_shifts(page):
    """page uint16 [64 tokens, 128 dims] -> int32 [128]."""
    e = ((page >> 7) & 0xff).astype(np.int64)
    cnt, tot = (e > 0).sum(axis=0), e.sum(axis=0)
    return np.where(cnt > 0, np.clip((tot + cnt // 2) // np.maximum(cnt, 1) - 125, -127, 127), 0).astype(np.int32)


def v_token_of():
    """[64 lanes, 128 values] -> token index of every value of a V page as its lanes see it."""
    lane = np.arange(64)[:, None]
    i = np.arange(128)[None, :]
    return 4 * (i >> 3) + 2 * (lane >> 5) + (i & 1)


def v_shifts(page):
    """page uint16 [64 lanes, 128 values] (V lane order) -> int32 [64] per token."""
    e = ((page >> 7) & 0xff).astype(np.int64)
    tok = v_token_of()
    tot = np.bincount(tok.ravel(), weights=e.ravel(), minlength=64).astype(np.int64)
    cnt = np.bincount(tok.ravel(), weights=(e > 0).ravel(), minlength=64).astype(np.int64)
    m = np.where(cnt > 0, (tot + cnt // 2) // np.maximum(cnt, 1), 0)
    ref = m[m > 0].min() if (m > 0).any() else 255
    return np.where(m > 0, np.minimum((m - ref) & ~1, 126), 0).astype(np.int32)


def seal(page, as_k=False):
    """page uint16 [64 lanes, 128 values] -> (sealed uint8 [13, 64, 16], fit bool [64]).  as_k: False / 0 plain, True / 1 the
    K form, 2 the V form."""
    v = page.astype(np.int64)
    out = np.zeros((13, 64, 16), dtype=np.uint8)
    unfit = np.zeros(64, dtype=bool)
    if as_k == 2:
        s = v_shifts(page)[v_token_of()].astype(np.int64)          # per value
        e = (v >> 7) & 0xff
        den = (e == 0) & ((v & 0x7f) != 0) & (s != 0)
        e2 = e - s
        bad = (e != 0) & np.where(e == 255, s != 0, e2 < 1)
        unfit = (den | bad).any(axis=1)
        v = np.where((e != 0) & ~bad, (v - (s << 7)) & 0xffff, v)
        out[12, :, 8] = v_shifts(page).astype(np.int64) & 0xff
    elif as_k:
        s = k_shifts(page)
        e = (v >> 7) & 0xff
        den = (e == 0) & ((v & 0x7f) != 0) & (s[None, :] != 0)
        e2 = e - s[None, :]
        bad = (e != 0) & np.where(e == 255, s[None, :] != 0, (e2 < 1) | (e2 > 254))
        unfit = (den | bad).any(axis=1)
        v = np.where((e != 0) & ~bad, (v - (s[None, :].astype(np.int64) << 7)) & 0xffff, v)
        sp = (s.astype(np.int64) & 0xff).reshape(64, 2)
        out[12, :, 8] = sp[:, 0]
        out[12, :, 9] = sp[:, 1]
    hb = (v >> 8) & 0x7f
    for l in range(64):
        d = np.unique(hb[l])
        if len(d) > 8 or unfit[l]:
            unfit[l] = True
            out[12, l, 12] = 1
            continue
        out[12, l, :len(d)] = d
        code = ((v[l] >> 12) & 8) | np.searchsorted(d, hb[l])
        low = (v[l] & 0xff).astype(np.uint8)
        out[0:8, l, :] = low.reshape(8, 16)
        c = code.reshape(16, 2, 4)                                   # [j][group][k]
        nib = (c[:, 0, :] | (c[:, 1, :] << 4)).astype(np.uint8)      # byte 4j + k
        out[8:12, l, :] = nib.reshape(4, 16)
    return out, ~unfit


def unseal(sealed, as_k=False):
    """sealed uint8 [13, 64, 16] -> (page uint16 [64, 128], fit bool [64]); unfit lanes come back as zeros."""
    low = sealed[0:8].transpose(1, 0, 2).reshape(64, 128).astype(np.int64)
    nib = sealed[8:12].transpose(1, 0, 2).reshape(64, 16, 4).astype(np.int64)
    code = np.concatenate([nib & 0xf, nib >> 4], axis=-1).reshape(64, 128)
    dic = sealed[12, :, :8].astype(np.int64)
    fit = sealed[12, :, 12:16].copy().view(np.uint32)[:, 0] == 0
    hi = np.take_along_axis(dic, code & 7, axis=-1)
    v = ((code >> 3) << 15) | (hi << 8) | low
    if as_k == 2:
        s = sealed[12, :, 8].astype(np.int64)[v_token_of()]
        v = np.where((v & 0x7fff) != 0, (v + (s << 7)) & 0xffff, v)
    elif as_k:
        s = sealed[12, :, 8:10].copy().view(np.int8).reshape(128).astype(np.int64)
        v = np.where((v & 0x7fff) != 0, (v + (s[None, :] << 7)) & 0xffff, v)
    return np.where(fit[:, None], v, 0).astype(np.uint16), fit
