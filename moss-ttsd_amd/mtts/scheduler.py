"""Continuous batching over one engine.

The reference serves a JSONL batch as one static batch: every row is stepped (and padded) until the longest
dialogue ends (modeling_asteroid.py:155-169).  Dialogues are independent, so here a finished dialogue leaves its
slot at once, its KV pages return to the engine's page pool, and the next queued one is prefilled into the slot while
the others keep decoding; the weight stream of every step is shared by whatever is resident.  Each dialogue's tokens
are exactly what it would get alone (`Engine.generate` with batch 1 and the same seed): nothing in a row's arithmetic
depends on its neighbours.

Admission is by free KV pages, not by slots alone: the pool (`Engine(kv_pool_pages=...)`) may hold far fewer pages
than `slots x max_seq_len` tokens.  A prompt is admitted when its pages are free with one page of headroom per
resident dialogue (each of them opens a new page every 64 steps); pages for generated tokens are taken as the
dialogues grow.  If the pool still runs dry mid-flight (`mtts_step` -> MTTS_ENOMEM) the dialogue with the fewest
generated rows is evicted and re-queued -- tokens are a function of (prompt, seed), so its re-run reproduces them --
and nothing new is admitted until a resident dialogue has finished (otherwise the evicted prompt would take the freed
pages straight back).
"""
from __future__ import annotations

import numpy as np

from . import capi


class ContinuousBatcher:
    def __init__(self, engine, slots, gen_cap, layers=None, do_samples=None, steps_per_poll=16):
        self.eng = engine
        self.slots = int(slots)
        self.gen_cap = int(gen_cap)
        self.steps_per_poll = int(steps_per_poll)
        self.evictions = 0
        engine.sched_open(self.slots, self.gen_cap, layers=layers, do_samples=do_samples)

    def run(self, prompts, max_new_tokens, seeds=None, base_seed=0, row_ids=None):
        """prompts: list of int64 [T_i,8] delay-shifted prompts (no padding); max_new_tokens: int or list
        (HF semantics: max_length = T_i + max_new).  seeds: one Philox key per dialogue (default base_seed + i, so
        that concurrent dialogues draw from different streams).  Returns a list of int64 [T_i-7+G_i, 8] in
        submission order.  row_ids: Philox row id per dialogue (default 0): with one shared seed and row_ids = the
        dialogues' positions in a batch they draw what that static batch's rows draw."""
        n = len(prompts)
        mnt = [max_new_tokens] * n if np.isscalar(max_new_tokens) else list(max_new_tokens)
        seeds = list(seeds) if seeds is not None else [int(base_seed) + i for i in range(n)]
        row_ids = [0] * n if row_ids is None else [int(r) for r in row_ids]
        results = [None] * n
        owner = [-1] * self.slots
        queue = list(range(n))
        steps = 0
        hold = False                                                  # after an eviction: wait for a dialogue to finish
        polls = 0
        while queue or any(o >= 0 for o in owner):
            polls += 1
            if polls > 1_000_000:
                raise capi.MttsError("continuous batcher made no progress (scheduling bug)")
            for s in range(self.slots):                               # refill free slots while pages last
                if owner[s] < 0 and queue and not hold:
                    i = queue[0]
                    ids = np.asarray(prompts[i], dtype=np.int64)
                    live = sum(1 for o in owner if o >= 0)
                    if live and self.eng.kv_pool_state()[1] < (ids.shape[0] - 7 + 64) // 64 + live + 1:
                        break                                         # not enough headroom next to the residents
                    try:
                        self.eng.submit(s, ids, ids.shape[0] + int(mnt[i]), seed=int(seeds[i]), row_id=row_ids[i])
                    except capi.MttsError as err:
                        if err.code != capi.ENOMEM:
                            raise
                        if not any(o >= 0 for o in owner):
                            raise                                     # does not fit an empty pool: not a scheduling matter
                        break                                         # wait for a resident dialogue to finish
                    queue.pop(0)
                    owner[s] = i
            try:
                self.eng.step(self.steps_per_poll)
                steps += self.steps_per_poll
            except capi.MttsError as err:
                if err.code != capi.ENOMEM:
                    raise
                st = self.eng.slot_states()                           # also returns the pages of finished slots
                live = [s for s in range(self.slots) if owner[s] >= 0 and st[s, 0]]
                if len(live) == sum(1 for s in range(self.slots) if owner[s] >= 0):
                    if len(live) <= 1:
                        raise                                         # one dialogue alone exhausts the pool
                    victim = min(live, key=lambda s: (st[s, 2], -s))  # least work lost
                    self.eng.evict(victim)
                    queue.insert(0, owner[victim])
                    owner[victim] = -1
                    self.evictions += 1
                    hold = True
            st = self.eng.slot_states()
            for s in range(self.slots):
                if owner[s] >= 0 and not st[s, 0]:                    # left the batch: collect
                    i = owner[s]
                    rows = self.eng.slot_read(s, self.gen_cap)
                    ids = np.asarray(prompts[i], dtype=np.int64)
                    results[i] = np.concatenate([ids[:ids.shape[0] - 7], rows], axis=0)
                    owner[s] = -1
                    hold = False
        self.engine_steps = steps
        return results
