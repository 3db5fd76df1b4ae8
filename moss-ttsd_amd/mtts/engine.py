"""Python handle on the HIP engine: device memory and streams come from
PyTorch-ROCm, everything that computes lives in libmtts.so."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import capi


def sampler_cfgs(layers=None, do_samples=None):
    """generation_config.layers / do_samples (reference modeling_asteroid.py:95-106) -> MttsSamplerCfg[8]."""
    arr = (capi.MttsSamplerCfg * 8)()
    for i in range(8):
        lc = (layers[i] if layers and i < len(layers) else {}) or {}
        s = arr[i]
        s.do_sample = 1 if (do_samples and do_samples[i]) else 0
        s.top_k = int(lc.get("top_k") or 0)
        tp = lc.get("top_p")
        s.top_p = float(tp) if tp is not None else 0.0
        s.one_minus_top_p = float(np.float32(1.0 - tp)) if tp is not None else 0.0
        t = lc.get("temperature")
        s.temperature = float(t) if t is not None else 0.0
        rp = lc.get("repetition_penalty")
        s.repetition_penalty = float(rp) if rp is not None else 0.0
    return arr


def rope_tables(head_dim, theta, rows, device, dtype=torch.bfloat16):
    """cos/sin exactly as Qwen3RotaryEmbedding.forward builds them (transformers
    modeling_qwen3.py:96-138): fp32 inv_freq, fp32 outer product, cos/sin, cast to the model dtype."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    pos = torch.arange(rows, dtype=torch.float)
    freqs = (inv_freq[:, None] @ pos[None, :]).transpose(0, 1)            # [rows, 64]
    return (freqs.cos().to(dtype).contiguous().to(device),
            freqs.sin().to(dtype).contiguous().to(device))


class Engine:
    def __init__(self, cfg: dict, max_batch: int, max_seq_len: int, device="cuda:0", kv_pool_pages: int = 0, dtype="bf16"):
        """dtype "bf16" (the reference default) or "fp32" (`inference.py --dtype fp32`: fp32 weights, arithmetic,
        K/V pages and logits -- the strict-parity mode, plain HBM-bound kernels)."""
        if dtype not in ("bf16", "fp32", "fp16"):
            raise NotImplementedError(f"dtype {dtype!r} is not built (bf16, fp32, fp16)")
        self.dtype = dtype
        # "fp16" (`inference.py --dtype fp16`): the fp32 engine with fp16 rounding points; its tensors are fp32 holding fp16 values
        self.tdtype = torch.bfloat16 if dtype == "bf16" else torch.float32
        self.model_dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(dtype, torch.float32)
        if not torch.cuda.is_available():
            raise capi.MttsError("no GPU visible: the mtts engine only runs on MI355X (no CPU fallback)")
        self.cfg = cfg
        self.device = torch.device(device)
        self.lib = capi.lib()
        c = capi.MttsConfig()
        for k in ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
                  "num_key_value_heads", "head_dim", "channels", "speech_vocab_size", "speech_pad_token",
                  "eos_token_id"):
            setattr(c, k, int(cfg[k]))
        c.speech_range_lo, c.speech_range_hi = int(cfg["speech_token_range"][0]), int(cfg["speech_token_range"][1])
        c.max_position = int(max_seq_len) + 24
        c.rms_norm_eps = float(cfg["rms_norm_eps"])
        c.max_batch, c.max_seq_len = int(max_batch), int(max_seq_len)
        c.kv_pool_pages = int(kv_pool_pages)          # 0: every slot can reach max_seq_len at once
        c.dtype = {"bf16": 0, "fp32": 1, "fp16": 2}[dtype]
        self._h = C.c_void_p()
        capi.check(self.lib.mtts_engine_create(C.byref(c), self.device.index or 0, C.byref(self._h)))
        cos, sin = (t.to(self.tdtype) for t in rope_tables(cfg["head_dim"], float(cfg["rope_theta"]), c.max_position,
                                                           self.device, self.model_dtype))
        torch.cuda.synchronize(self.device)
        capi.check(self.lib.mtts_bind_rope(self._h, cos.data_ptr(), sin.data_ptr(), c.max_position, None))
        torch.cuda.synchronize(self.device)

    def close(self):
        if self._h:
            self.lib.mtts_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights -----------------------------------------------------------
    def bind(self, name: str, tensor: torch.Tensor):
        t = tensor.to(device=self.device).to(self.model_dtype).to(self.tdtype).contiguous()      # (cast like model.to(dtype) casts)
        rows, cols = (t.shape[0], t.shape[1]) if t.dim() == 2 else (t.shape[0], 1)
        capi.check(self.lib.mtts_bind_weight(self._h, name.encode(), t.data_ptr(), rows, cols, None))
        torch.cuda.synchronize(self.device)     # the engine has packed its own copy; `t` may go

    def bind_state_dict(self, sd):
        for k, v in sd.items():
            if k.startswith("lm_heads.") or k.endswith("embed_tokens.weight"):
                continue
            if isinstance(v, np.ndarray):
                v = torch.from_numpy(v)
            self.bind(k, v)
        capi.check(self.lib.mtts_weights_ready(self._h))

    # ---- generation --------------------------------------------------------
    @staticmethod
    def _host_inputs(input_ids, attention_mask):
        ids = np.ascontiguousarray(np.asarray(input_ids.cpu() if torch.is_tensor(input_ids) else input_ids), dtype=np.int64)
        m = attention_mask.cpu().numpy() if torch.is_tensor(attention_mask) else np.asarray(attention_mask)
        m = np.ascontiguousarray((m > 0).astype(np.uint8))
        assert ids.ndim == 3 and ids.shape[2] == 8 and m.shape == ids.shape[:2]
        return ids, m

    def _set_row_ids(self, row_ids, B):
        if row_ids is not None:
            r = np.ascontiguousarray(row_ids, dtype=np.int32)
            assert r.shape == (B,)
            capi.check(self.lib.mtts_set_row_ids(self._h, r.ctypes.data, B))

    def generate(self, input_ids, attention_mask, max_length, layers=None, do_samples=None, seed=0, forced=None,
                 forced_as_draw=False, row_ids=None):
        """forced (verification hook): int64 [B,G,8] full sequences of a reference run.  -> (ids, decisions [steps,B,8]).
        forced_as_draw: the forced row replaces each step's raw draw before the state machine (replay of a SAMPLED
        run); decisions are then the raw draws.  forced_as_draw="all": also for rows that max_length has cut off (the
        reference keeps evaluating them; scripted tests of chained resurrections)."""
        ids, m = self._host_inputs(input_ids, attention_mask)
        B, T, _ = ids.shape
        cap = int(max_length) + 6 * B + 8  # flushes that start at / run past max_length, chained (include/mtts.h: mtts_generate)
        self._B = B
        mode = 0 if (forced is None or not forced_as_draw) else (2 if forced_as_draw == "all" else 1)
        capi.check(self.lib.mtts_set_forced_mode(self._h, mode))
        out = np.zeros((B, cap, 8), dtype=np.int64)
        out_len = C.c_int32(0)
        scfg = sampler_cfgs(layers, do_samples)
        self._set_row_ids(row_ids, B)       # Philox row id of each row (default: its position in this batch)
        fptr, flen, dptr, dec = None, 0, None, None
        if forced is not None:
            forced = np.ascontiguousarray(forced, dtype=np.int64)
            flen = forced.shape[1]
            fptr = forced.ctypes.data
            dec = np.zeros((cap, B, 8), dtype=np.int64)
            dptr = dec.ctypes.data
        capi.check(self.lib.mtts_generate(self._h, ids.ctypes.data, m.ctypes.data, B, T, int(max_length), scfg,
                                          C.c_uint64(seed), out.ctypes.data, cap, C.byref(out_len), fptr, flen, dptr, None))
        res = out[:, :out_len.value].copy()
        if forced is not None:
            return res, dec[:out_len.value - (T - 7)].copy()
        return res

    def begin(self, input_ids, attention_mask, max_length, layers=None, do_samples=None, seed=0, row_ids=None):
        ids, m = self._host_inputs(input_ids, attention_mask)
        B, T, _ = ids.shape
        self._B, self._T = B, T
        self._set_row_ids(row_ids, B)
        capi.check(self.lib.mtts_begin(self._h, ids.ctypes.data, m.ctypes.data, B, T, int(max_length),
                                       sampler_cfgs(layers, do_samples), C.c_uint64(seed), None))

    def step(self, n=1, stream=None):
        """Issue n decode steps on `stream` (torch.cuda.Stream; default: the device's default stream)."""
        capi.check(self.lib.mtts_step(self._h, int(n), C.c_void_p(stream.cuda_stream) if stream is not None else None))

    def sync_state(self, stream=None):
        s, d = C.c_int32(0), C.c_int32(0)
        capi.check(self.lib.mtts_sync_state(self._h, C.byref(s), C.byref(d),
                                            C.c_void_p(stream.cuda_stream) if stream is not None else None))
        return s.value, bool(d.value)

    def read_generated(self, capacity_steps):
        buf = np.zeros((capacity_steps, self._B, 8), dtype=np.int64)
        n = C.c_int32(0)
        capi.check(self.lib.mtts_read_generated(self._h, buf.ctypes.data, capacity_steps, C.byref(n)))
        return buf[:n.value]

    def read_logits(self):
        V0, Vs = self.cfg["vocab_size"], self.cfg["speech_vocab_size"]
        if self.dtype != "bf16":
            l0 = np.zeros((self._B, V0), dtype=np.float32)
            l17 = np.zeros((7, self._B, Vs), dtype=np.float32)
            capi.check(self.lib.mtts_read_logits_f32(self._h, l0.ctypes.data, l17.ctypes.data, None))
            return l0, l17
        l0 = np.zeros((self._B, V0), dtype=np.uint16)
        l17 = np.zeros((7, self._B, Vs), dtype=np.uint16)
        capi.check(self.lib.mtts_read_logits(self._h, l0.ctypes.data, l17.ctypes.data, None))
        f = lambda u: (u.astype(np.uint32) << 16).view(np.float32)
        return f(l0), f(l17)

    # ---- continuous batching ---------------------------------------------------------
    def sched_open(self, slots, gen_cap, layers=None, do_samples=None):
        self._B = int(slots)
        capi.check(self.lib.mtts_sched_open(self._h, int(slots), int(gen_cap), sampler_cfgs(layers, do_samples), None))

    def submit(self, slot, ids, max_length, seed=0, row_id=0):
        """ids int64 [T,8]: one delay-shifted prompt without padding (what shifting_inputs returns).  The dialogue
        draws from the Philox stream (seed; step, row_id, channel)."""
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        assert ids.ndim == 2 and ids.shape[1] == 8
        capi.check(self.lib.mtts_slot_submit_row(self._h, int(slot), ids.ctypes.data, ids.shape[0], int(max_length),
                                                 C.c_uint64(seed), int(row_id), None))

    def slot_states(self):
        """-> int32 [slots,4]: active, unfinished, rows generated, tokens cached."""
        st = np.zeros((self._B, 4), dtype=np.int32)
        capi.check(self.lib.mtts_slot_states(self._h, st.ctypes.data, None))
        return st

    def slot_read(self, slot, capacity):
        buf = np.zeros((int(capacity), 8), dtype=np.int64)
        n = C.c_int32(0)
        capi.check(self.lib.mtts_slot_read(self._h, int(slot), buf.ctypes.data, int(capacity), C.byref(n)))
        return buf[:n.value].copy()

    def evict(self, slot):
        """Scheduler mode: drop the dialogue in `slot` and return its KV pages (it is re-submitted later)."""
        capi.check(self.lib.mtts_slot_evict(self._h, int(slot), None))

    def kv_pool_state(self):
        """-> (pages in the pool, pages free, pages per page-table row)."""
        a, b, c = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        capi.check(self.lib.mtts_kv_pool_state(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def page_table(self, max_batch):
        """-> (int32 [max_batch, pages per row] page table as the host tracks it, int32 [max_batch] pages owned)."""
        _, _, mp = self.kv_pool_state()
        t = np.zeros((int(max_batch), mp), dtype=np.int32)
        n = np.zeros(int(max_batch), dtype=np.int32)
        capi.check(self.lib.mtts_read_page_table(self._h, t.ctypes.data, n.ctypes.data))
        return t, n

    def device_page_table(self, max_batch):
        """The page table as the device holds it (verification hook) -> int32 [max_batch, pages per row]."""
        _, _, mp = self.kv_pool_state()
        t = np.zeros((int(max_batch), mp), dtype=np.int32)
        capi.check(self.lib.mtts_debug_read_device_page_table(self._h, t.ctypes.data, None))
        return t

    def seq_state(self):
        """-> (needs_additional_steps[B], unfinished[B], kv_len[B]) numpy int32."""
        a, u, k = (np.zeros(self._B, dtype=np.int32) for _ in range(3))
        capi.check(self.lib.mtts_read_seq_state(self._h, a.ctypes.data, u.ctypes.data, k.ctypes.data, None))
        return a, u, k

    def export_codes(self, first, n, stream=None):
        """Frames first..first+n-1 -> int64 [8,B,n] device tensor, enqueued on `stream` (torch.cuda.Stream or None)."""
        out = torch.empty(8, self._B, n, dtype=torch.int64, device=self.device)
        sp = C.c_void_p(stream.cuda_stream) if stream is not None else None
        capi.check(self.lib.mtts_export_codes(self._h, int(first), int(n), out.data_ptr(), sp))
        return out

    def kv_pack_stats(self):
        """Sealed KV pages of the live dialogues -> dict(k_pages, k_unsealed, v_pages, v_unsealed, k_layers_on, v_layers_on)
        (page x kv head x layer counts; `unsealed` = a lane did not fit the 13-bit form, the page is read as bf16;
        `*_layers_on` = layers whose reads currently take the sealed pages)."""
        o = np.zeros(6, dtype=np.int64)
        capi.check(self.lib.mtts_debug_kv_pack_stats(self._h, o.ctypes.data))
        return dict(zip(("k_pages", "k_unsealed", "v_pages", "v_unsealed", "k_layers_on", "v_layers_on"), (int(x) for x in o)))

    def debug_set_kv_len(self, n):
        capi.check(self.lib.mtts_debug_set_kv_len(self._h, int(n)))

    def attn_bench(self, phase, iters=112):
        """-> (avg ms per launch, algorithmic bytes per launch) of attention pass `phase` (1 scores, 2 PV) at the
        current decode state, measured over a train of back-to-back launches."""
        ms, by = C.c_float(0), C.c_int64(0)
        capi.check(self.lib.mtts_k_attn_bench(self._h, int(phase), int(iters), C.byref(ms), C.byref(by)))
        return ms.value, by.value

    def profile(self, on=True):
        capi.check(self.lib.mtts_profile_enable(self._h, 1 if on else 0))

    def profile_read(self, which):
        ms, n, by = C.c_double(0), C.c_int64(0), C.c_int64(0)
        capi.check(self.lib.mtts_profile_read(self._h, which, C.byref(ms), C.byref(n), C.byref(by)))
        return ms.value, n.value, by.value
