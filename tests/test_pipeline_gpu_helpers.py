"""Shared by the pipeline tests: the yaml-shaped `generator_params` of a (reduced-depth) XY_Tokenizer config."""


def generator_params(cfg):
    enc = {"encoder_layers": cfg["enc_layers"], "d_model": 768, "encoder_attention_heads": 12, "encoder_ffn_dim": 3072,
           "max_audio_seconds": 30, "sampling_rate": 16000, "hop_length": 160, "stride_size": 2}
    return {"input_sample_rate": 16000, "output_sample_rate": 24000,
            "feature_extractor_kwargs": {"n_fft": 400, "hop_length": 160, "nb_max_frames": 3000},
            "semantic_encoder_kwargs": enc, "acoustic_encoder_kwargs": enc,
            "semantic_encoder_adapter_kwargs": {"encoder_layers": cfg["sem_adapter_layers"]},
            "pre_rvq_adapter_kwargs": {"encoder_layers": cfg["pre_rvq_layers"]},
            "downsample_kwargs": {"avg_pooler": 4},
            "quantizer_kwargs": {"num_quantizers": 8, "codebook_size": 1024, "rvq_dim": 512, "output_dim": 3072},
            "post_rvq_adapter_kwargs": {"encoder_layers": cfg["adapter_layers"], "d_model": 768,
                                        "encoder_attention_heads": 12, "encoder_ffn_dim": 3072, "max_source_positions": 375},
            "upsample_kwargs": {"stride": 4},
            "acoustic_decoder_kwargs": {"decoder_layers": cfg["dec_layers"], "d_model": 768, "decoder_attention_heads": 12,
                                        "decoder_ffn_dim": 3072, "max_audio_seconds": 30, "sampling_rate": 16000,
                                        "hop_length": 160, "stride_size": 2, "num_mel_bins": 80},
            "vocos_kwargs": {"dim": 512, "intermediate_dim": 4096, "num_layers": cfg["voc_layers"], "n_fft": 960, "hop_size": 240}}
