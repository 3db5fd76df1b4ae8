"""Codec decoder time per 30 s window against windows per call (tuning aid, GPU box only): 18.4 / 11.1 / 7.6 / 6.3 / 6.1 / 6.0 ms
for 1 / 2 / 4 / 8 / 16 / 32 windows."""
import json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "moss-ttsd_amd")); sys.path.insert(0, ROOT)
import torch, bench
for w in (1, 2, 4, 8, 16, 32):
    r = bench.codec_leg(torch.device("cuda:0"), windows=w)
    print(w, round(r["ms_per_window"], 3), flush=True)
