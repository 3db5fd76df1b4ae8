"""ctypes binding of libmtts.so (include/mtts.h).  Fails loudly when the HIP
library is missing: there is no CPU fallback in the product path."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libmtts.so")


class MttsConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
        "num_key_value_heads", "head_dim", "channels", "speech_vocab_size", "speech_pad_token",
        "speech_range_lo", "speech_range_hi", "eos_token_id", "max_position")] + [
        ("rms_norm_eps", C.c_float), ("max_batch", C.c_int32), ("max_seq_len", C.c_int32), ("kv_pool_pages", C.c_int32), ("dtype", C.c_int32)]


class MttsSamplerCfg(C.Structure):
    _fields_ = [("do_sample", C.c_int32), ("top_k", C.c_int32), ("top_p", C.c_float),
                ("one_minus_top_p", C.c_float), ("temperature", C.c_float), ("repetition_penalty", C.c_float)]


class MttsCodecConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "nq", "codebook_size", "rvq_dim", "quant_out_dim",
        "adapter_layers", "adapter_dim", "adapter_heads", "adapter_ffn", "adapter_max_pos", "up_stride",
        "dec_layers", "dec_dim", "dec_heads", "dec_ffn", "dec_max_pos", "mel_bins",
        "voc_dim", "voc_inter", "voc_layers", "n_fft", "hop",
        "mel_n_fft", "mel_hop", "mel_frames", "enc_layers", "enc_dim", "enc_heads", "enc_ffn", "enc_max_pos",
        "sem_adapter_layers", "pre_rvq_layers", "down_pool")]


class MttsError(RuntimeError):
    def __init__(self, msg, code=0):
        super().__init__(msg)
        self.code = code


ENOMEM = -3


_lib = None

_SIGS = {
    "mtts_last_error": (C.c_char_p, []),
    "mtts_version": (C.c_int32, []),
    "mtts_engine_create": (C.c_int32, [C.POINTER(MttsConfig), C.c_int32, C.POINTER(C.c_void_p)]),
    "mtts_engine_destroy": (C.c_int32, [C.c_void_p]),
    "mtts_bind_weight": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "mtts_bind_rope": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "mtts_weights_ready": (C.c_int32, [C.c_void_p]),
    "mtts_generate": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                  C.POINTER(MttsSamplerCfg), C.c_uint64, C.c_void_p, C.c_int32,
                                  C.POINTER(C.c_int32), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "mtts_begin": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                               C.POINTER(MttsSamplerCfg), C.c_uint64, C.c_void_p]),
    "mtts_step": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p]),
    "mtts_sync_state": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p]),
    "mtts_read_generated": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "mtts_read_logits": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mtts_read_logits_f32": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mtts_sched_open": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(MttsSamplerCfg), C.c_void_p]),
    "mtts_slot_submit": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_uint64, C.c_void_p]),
    "mtts_slot_submit_row": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_uint64, C.c_int32, C.c_void_p]),
    "mtts_set_row_ids": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32]),
    "mtts_slot_states": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mtts_slot_read": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "mtts_read_seq_state": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mtts_slot_evict": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p]),
    "mtts_kv_pool_state": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mtts_read_page_table": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mtts_set_forced_mode": (C.c_int32, [C.c_void_p, C.c_int32]),
    "mtts_debug_read_device_page_table": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mtts_export_codes": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "mtts_profile_enable": (C.c_int32, [C.c_void_p, C.c_int32]),
    "mtts_profile_read": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int64)]),
    "mtts_debug_set_kv_len": (C.c_int32, [C.c_void_p, C.c_int32]),
    "mtts_debug_kv_pack_stats": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "mtts_k_kv_seal": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "mtts_k_attn_bench": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_int64)]),
    "mtts_k_gemm_bench": (C.c_int32, [C.c_int32] * 7 + [C.POINTER(C.c_float)]),
    "mtts_k_gemm_bf16": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_void_p]),
    "mtts_k_rmsnorm": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "mtts_k_rope_kvwrite": (C.c_int32, [C.c_void_p] * 6 + [C.c_int32, C.c_int32, C.c_int32, C.c_float] + [C.c_void_p] * 4),
    "mtts_k_paged_attn_decode": (C.c_int32, [C.c_void_p] * 5 + [C.c_int32] * 4 + [C.c_void_p, C.c_void_p]),
    "mtts_k_sample": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(MttsSamplerCfg),
                                  C.c_int32, C.c_uint64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "mtts_codec_last_error": (C.c_char_p, []),
    "mtts_codec_create": (C.c_int32, [C.POINTER(MttsCodecConfig), C.c_int32, C.POINTER(C.c_void_p)]),
    "mtts_codec_destroy": (C.c_int32, [C.c_void_p]),
    "mtts_codec_bind": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "mtts_codec_detokenize": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "mtts_codec_detokenize_async": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "mtts_codec_check": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "mtts_codec_tokenize": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mtts_k_gemm_f32": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_void_p]),
    "mtts_k_gemm_planes_bench": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                             C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
}


def exported_symbols():
    return sorted(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MttsError(f"{LIB_PATH} is missing: build it with `python moss-ttsd_amd/build.py` "
                            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise MttsError(f"libmtts error {rc}: {lib().mtts_last_error().decode()}", rc)
