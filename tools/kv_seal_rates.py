"""Which K / V statistics seal?  Seal rate (share of 64-token pages whose every lane fits the 13-bit form) of synthetic pages
under the distributions real checkpoints may have, with oracle/kv_seal_oracle.py (CPU only; a deployer with a checkpoint runs
mtts_debug_kv_pack_stats instead).  K page = [64 tokens][128 dims] (lane = token, rescaled per dim);
V page as a lane sees it = [64 lanes][128 values], lane = 4 dims x the 32 tokens of one pair parity (rescaled per token)."""
import json
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import kv_seal_oracle as ks


def bits(x):
    return (np.ascontiguousarray(x, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


def v_lanes(x):                      # x [64 tokens, 128 dims] -> the V page as its lanes see it (attn.hip: seal_lane_v)
    return x.reshape(16, 2, 2, 32, 4).transpose(1, 3, 0, 4, 2).reshape(64, 128)   # lane 32 sub + dl, value 8 it + 2 c + h


def rates(make, pages=60, seed=0):
    rng = np.random.default_rng(seed)
    k = v = 0
    for _ in range(pages):
        x = make(rng)
        k += ks.seal(bits(x), as_k=True)[1].all()
        v += ks.seal(bits(v_lanes(x)), as_k=2)[1].all()
    return round(k / pages, 3), round(v / pages, 3)


def gauss(rng): return rng.standard_normal((64, 128))
def dim_scales(sig): return lambda rng: rng.standard_normal((64, 128)) * np.exp(rng.standard_normal((1, 128)) * sig)
def token_scales(sig): return lambda rng: rng.standard_normal((64, 128)) * np.exp(rng.standard_normal((64, 1)) * sig)
def student(nu): return lambda rng: rng.standard_t(nu, (64, 128))
def outlier_dims(rng):
    x = rng.standard_normal((64, 128)); x[:, rng.choice(128, 4, replace=False)] *= 50.0; return x
def sink_token(rng):
    x = rng.standard_normal((64, 128)); x[0] *= 30.0; return x


cases = {"gaussian": gauss, "per-dim scale, log-normal sigma 0.5": dim_scales(0.5), "sigma 1.0": dim_scales(1.0), "sigma 2.0": dim_scales(2.0),
         "per-token scale, sigma 0.5": token_scales(0.5), "per-token sigma 1.0": token_scales(1.0), "student-t nu=4": student(4), "student-t nu=2.5": student(2.5),
         "4 outlier dims x50": outlier_dims, "one token x30 (sink)": sink_token}
out = {name: dict(zip(("k_pages_sealed", "v_pages_sealed"), rates(f))) for name, f in cases.items()}
print(json.dumps(out, indent=1))
