"""End-to-end demo on synthetic weights (no checkpoint exists in this environment):
JSONL items -> generation_utils.process_batch (MI355X model + codec) -> PCM16 wav files.

    python tools/demo_synthetic.py --out gpurun_out/demo

With a real checkpoint the reference's own inference.py runs unchanged:
    PYTHONPATH=moss-ttsd_amd python /path/to/MOSS-TTSD/inference.py --jsonl examples/examples.jsonl
"""
import argparse
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "moss-ttsd_amd"))
import torch  # noqa: E402
import generation_utils as gu  # noqa: E402
from modeling_asteroid import AsteroidTTSInstruct, GenerationConfig  # noqa: E402
from XY_Tokenizer.xy_tokenizer.model import XY_Tokenizer  # noqa: E402
from mtts import synth, synth_codec  # noqa: E402


class CharTokenizer:
    """Stand-in for the HF tokenizer (only .encode and .pad_token_id are used by the pipeline)."""
    pad_token_id = 151643

    def encode(self, s):
        return [min(ord(c), 151000) for c in s]


def generator_params(c):
    enc = {"encoder_layers": c["enc_layers"], "d_model": 768, "encoder_attention_heads": 12, "encoder_ffn_dim": 3072,
           "max_audio_seconds": 30, "sampling_rate": 16000, "hop_length": 160, "stride_size": 2}
    return {"input_sample_rate": 16000, "output_sample_rate": 24000,
            "feature_extractor_kwargs": {"n_fft": 400, "hop_length": 160, "nb_max_frames": 3000},
            "semantic_encoder_kwargs": enc, "acoustic_encoder_kwargs": enc,
            "semantic_encoder_adapter_kwargs": {"encoder_layers": c["sem_adapter_layers"]},
            "pre_rvq_adapter_kwargs": {"encoder_layers": c["pre_rvq_layers"]}, "downsample_kwargs": {"avg_pooler": 4},
            "quantizer_kwargs": {"num_quantizers": 8, "codebook_size": 1024, "rvq_dim": 512, "output_dim": 3072},
            "post_rvq_adapter_kwargs": {"encoder_layers": c["adapter_layers"], "d_model": 768, "encoder_attention_heads": 12,
                                        "encoder_ffn_dim": 3072, "max_source_positions": 375},
            "upsample_kwargs": {"stride": 4},
            "acoustic_decoder_kwargs": {"decoder_layers": c["dec_layers"], "d_model": 768, "decoder_attention_heads": 12,
                                        "decoder_ffn_dim": 3072, "max_audio_seconds": 30, "sampling_rate": 16000,
                                        "hop_length": 160, "stride_size": 2, "num_mel_bins": 80},
            "vocos_kwargs": {"dim": 512, "intermediate_dim": 4096, "num_layers": c["voc_layers"], "n_fft": 960, "hop_size": 240}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="outputs")
    ap.add_argument("--max-new-tokens", type=int, default=60)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    cfg = synth.tiny()
    w = synth.synth_weights(cfg, 1, emb_row_sigma=0.3, speech_boost=10.0, eos_boost=1.0)   # channel 0 stays in the speech range
    gen_cfg = GenerationConfig(max_new_tokens=args.max_new_tokens, eos_token_id=cfg["eos_token_id"], do_samples=[True] * 8,
                               layers=[dict(top_k=50, top_p=0.95, temperature=1.0, repetition_penalty=1.1)] * 8)
    model = AsteroidTTSInstruct.from_state_dict(cfg, w, gen_cfg).eval().to("cuda")
    ccfg = synth_codec.reduced()
    spt = XY_Tokenizer(generator_params(ccfg), synth_codec.synth_weights(ccfg, 2, encoder=True)).eval().to("cuda")
    prompt = torch.from_numpy(synth_codec.synth_wavs(3, [16000 * 3])[0])[None]
    items = [{"text": "[S1]Hello, this is the MI355X engine.[S2]And this is speaker two."},
             {"text": "[S1]A cloned voice.", "prompt_audio": (prompt, 16000), "prompt_text": "[S1]reference audio"}]
    texts, results = gu.process_batch(items, CharTokenizer(), model, spt, "cuda", "You are a speech synthesizer.", 0,
                                      use_normalize=True)
    for i, r in enumerate(results):
        if r is None:
            print(f"sample {i}: failed")
            continue
        path = os.path.join(args.out, f"output_{i}.wav")
        gu.save_wav(path, r["audio_data"], r["sample_rate"])
        print(f"sample {i}: {r['audio_data'].shape[1] / r['sample_rate']:.2f} s -> {path}")


if __name__ == "__main__":
    main()
