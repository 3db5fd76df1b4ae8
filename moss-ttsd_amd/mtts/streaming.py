"""Codec decode overlapped with the autoregressive loop.

The decode loop is HBM-bound, the codec decoder is fp32-MFMA-bound: on two HIP streams they share
the GPU.  As soon as a 30 s window (375 codes) of a still-running dialogue is complete it is
decoded on a side stream while the main stream keeps generating; only each dialogue's last,
shorter window waits for the end.  Results are bit-identical to decoding after the loop, window
by window, as `generation_utils.process_batch` does (every sample is decoded on its own there:
reference generation_utils.py:434-450, XY_Tokenizer/xy_tokenizer/model.py:195-256).
"""
from __future__ import annotations

import numpy as np
import torch

CHUNK = 375          # codes per window (30 s)
STRIDE = 250         # codes kept per window (20 s, overlap_seconds=10)
UP = 1920            # samples per code


def valid_lengths(gen, speech_pad=1024):
    """gen int64 [G,B,8] -> per-row number of valid frames (last frame whose channel 1 is not the pad id, +1;
    reference find_max_valid_positions, generation_utils.py:240-249)."""
    G, B, _ = gen.shape
    n = G - 7
    if n <= 0:
        return np.zeros(B, dtype=np.int64)
    ch1 = gen[1:n + 1, :, 1]                       # un-shifted channel 1
    valid = ch1 != speech_pad
    last = n - 1 - np.argmax(valid[::-1], axis=0)
    return np.where(valid.any(axis=0), last + 1, 0)


def generate_with_overlapped_decode(engine, codec, input_ids, attention_mask, max_length, layers=None,
                                    do_samples=None, seed=0, steps_per_round=STRIDE):
    """-> (generated rows int64 [G,B,8], wavs: list of B fp32 device tensors)"""
    dev = engine.device
    side = torch.cuda.Stream(device=dev)
    main = torch.cuda.default_stream(dev)
    engine.begin(input_ids, attention_mask, max_length, layers=layers, do_samples=do_samples, seed=seed)
    B = engine._B
    max_steps = int(max_length) - (np.asarray(input_ids).shape[1] - 7) + 14     # + flushes that start at / run past max_length
    done_windows = [dict() for _ in range(B)]          # row -> {window k: wav tensor [<=480000]}
    keep = []                                          # tensors that must outlive the side stream's use
    next_win = 0
    issued = 0
    fin = False
    while not fin and issued < max_steps:
        n = min(steps_per_round, max_steps - issued)
        engine.step(n)
        issued += n
        st, fin = engine.sync_state()
        _, unfinished, _ = engine.seq_state()
        while STRIDE * next_win + CHUNK + 7 <= st:
            rows = [b for b in range(B) if unfinished[b]]           # still running => longer than this window
            if rows:
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    codes = engine.export_codes(STRIDE * next_win, CHUNK, stream=side)
                    sel = codes[:, rows].contiguous() if len(rows) < B else codes
                    wav = codec.detokenize_async(sel, [CHUNK] * len(rows), side)
                keep += [codes, sel, wav]
                for j, b in enumerate(rows):
                    done_windows[b][next_win] = wav[j, :STRIDE * UP]
            next_win += 1
    steps, _ = engine.sync_state()
    gen = engine.read_generated(steps)
    codec.check(side)
    lens = valid_lengths(gen, engine.cfg["speech_pad_token"])
    # tails (and anything a finished row missed): rows whose remaining window has the same extent are decoded in
    # one call (equal T, no padding => identical to the reference's per-sample decode)
    n_frames = gen.shape[0] - 7
    todo = {}
    for b in range(B):
        L = int(lens[b])
        k = 0
        while STRIDE * k < L:
            start = STRIDE * k
            if not (k in done_windows[b] and start + CHUNK <= L):
                todo.setdefault((start, min(start + CHUNK, L)), []).append(b)
            k += 1
    if todo:
        full = engine.export_codes(0, n_frames)
        torch.cuda.synchronize(dev)
        for (start, end), rows in todo.items():
            w = codec.detokenize(full[:, rows, start:end].contiguous(), [end - start] * len(rows))
            for j, b in enumerate(rows):
                done_windows[b][start // STRIDE] = w[j]
    wavs = []
    for b in range(B):
        L = int(lens[b])
        parts = [done_windows[b][k][:min(L - STRIDE * k, STRIDE) * UP] for k in range((L + STRIDE - 1) // STRIDE)]
        wavs.append(torch.cat(parts) if parts else torch.zeros(0, device=dev))
    return gen, wavs
