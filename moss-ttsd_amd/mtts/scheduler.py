"""Continuous batching over one engine.

The reference serves a JSONL batch as one static batch: every row is stepped (and padded) until the longest
dialogue ends (modeling_asteroid.py:155-169).  Dialogues are independent, so here a finished dialogue leaves its
slot at once and the next queued one is prefilled into it while the others keep decoding; the weight stream of
every step is shared by whatever is resident.  Each dialogue's tokens are exactly what it would get alone
(`Engine.generate` with batch 1 and the same seed): nothing in a row's arithmetic depends on its neighbours.
"""
from __future__ import annotations

import numpy as np


class ContinuousBatcher:
    def __init__(self, engine, slots, gen_cap, layers=None, do_samples=None, steps_per_poll=16):
        self.eng = engine
        self.slots = int(slots)
        self.gen_cap = int(gen_cap)
        self.steps_per_poll = int(steps_per_poll)
        engine.sched_open(self.slots, self.gen_cap, layers=layers, do_samples=do_samples)

    def run(self, prompts, max_new_tokens, seeds=None):
        """prompts: list of int64 [T_i,8] delay-shifted prompts (no padding); max_new_tokens: int or list
        (HF semantics: max_length = T_i + max_new).  Returns a list of int64 [T_i-7+G_i, 8] in submission order."""
        n = len(prompts)
        mnt = [max_new_tokens] * n if np.isscalar(max_new_tokens) else list(max_new_tokens)
        seeds = list(seeds) if seeds is not None else [0] * n
        results = [None] * n
        owner = [-1] * self.slots
        queue = list(range(n))
        steps = 0
        while queue or any(o >= 0 for o in owner):
            for s in range(self.slots):                               # refill free slots
                if owner[s] < 0 and queue:
                    i = queue.pop(0)
                    ids = np.asarray(prompts[i], dtype=np.int64)
                    self.eng.submit(s, ids, ids.shape[0] + int(mnt[i]), seed=int(seeds[i]))
                    owner[s] = i
            self.eng.step(self.steps_per_poll)
            steps += self.steps_per_poll
            st = self.eng.slot_states()
            for s in range(self.slots):
                if owner[s] >= 0 and not st[s, 0]:                    # left the batch: collect
                    i = owner[s]
                    rows = self.eng.slot_read(s, self.gen_cap)
                    ids = np.asarray(prompts[i], dtype=np.int64)
                    results[i] = np.concatenate([ids[:ids.shape[0] - 7], rows], axis=0)
                    owner[s] = -1
        self.engine_steps = steps
        return results
