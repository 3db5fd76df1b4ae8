"""Python handle on the HIP codec decoder (csrc/codec.hip) + the loader that turns a
reference-format XY_Tokenizer state dict into the engine's role tensors."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import capi


MttsCodecConfig = capi.MttsCodecConfig


def _check(rc):
    if rc != 0:
        raise capi.MttsError(f"libmtts codec error {rc}: {capi.lib().mtts_codec_last_error().decode()}")


def sinusoids(length, channels, max_timescale=10000):
    """reference nn/modules.py:25-31, verbatim arithmetic (torch fp32)."""
    inc = np.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    st = torch.arange(length)[:, None] * inv[None, :]
    return torch.cat([torch.sin(st), torch.cos(st)], dim=1)


def istft_basis(n_fft):
    """irfft(n=n_fft, norm='backward') as a real matrix: frames = [Re | Im] @ basis, rows padded to 16."""
    nb = n_fft // 2 + 1
    k = np.arange(nb, dtype=np.float64)[:, None]
    n = np.arange(n_fft, dtype=np.float64)[None, :]
    ck = np.full((nb, 1), 2.0)
    ck[0] = 1.0
    ck[-1] = 1.0
    ang = 2.0 * np.pi * k * n / n_fft
    re = ck * np.cos(ang) / n_fft
    im = -ck * np.sin(ang) / n_fft          # irfft ignores Im of DC and Nyquist: sin() is 0 there anyway
    im[0] = 0.0
    im[-1] = 0.0
    ld = (2 * nb + 15) // 16 * 16
    out = np.zeros((ld, n_fft), dtype=np.float32)
    out[:nb] = re
    out[nb:2 * nb] = im
    return torch.from_numpy(out)


def mel_filter_bank_slaney(n_freq, n_mels, sr, fmin, fmax):
    """transformers.audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney") (third-party), which the
    reference's MelFeatureExtractor uses (nn/feature_extractor.py:46-54): triangular filters on the slaney
    mel scale, area-normalised.  -> [n_freq, n_mels] fp32."""
    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * (27.0 / np.log(6.4)), 3.0 * f / 200.0)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), 200.0 * m / 3.0)

    fft_freqs = np.linspace(0, sr // 2, n_freq)
    filt = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(filt)
    slopes = filt[None, :] - fft_freqs[:, None]
    fb = np.maximum(0.0, np.minimum(-slopes[:, :-2] / fdiff[:-1], slopes[:, 2:] / fdiff[1:]))
    fb *= (2.0 / (filt[2:n_mels + 2] - filt[:n_mels]))[None, :]
    return fb.astype(np.float32)


def encoder_role_tensors(cfg, sd):
    """Encode-side roles."""
    t = lambda k: (torch.from_numpy(sd[k]) if isinstance(sd[k], np.ndarray) else sd[k]).detach().float().cpu()
    r = {}
    n_fft = cfg["mel_n_fft"]
    nb = n_fft // 2 + 1
    ldri, ldp = (2 * nb + 15) // 16 * 16, (nb + 15) // 16 * 16
    n = np.arange(n_fft, dtype=np.float64)[:, None]
    kk = np.arange(nb, dtype=np.float64)[None, :]
    dft = np.zeros((n_fft, ldri), dtype=np.float32)
    dft[:, :nb] = np.cos(2 * np.pi * n * kk / n_fft)
    dft[:, nb:2 * nb] = -np.sin(2 * np.pi * n * kk / n_fft)
    r["mel.dft"] = torch.from_numpy(dft)
    fb = np.zeros((ldp, cfg["mel_bins"]), dtype=np.float32)
    fb[:nb] = mel_filter_bank_slaney(nb, cfg["mel_bins"], cfg["input_sample_rate"], 0.0, cfg["input_sample_rate"] / 2)
    r["mel.fb"] = torch.from_numpy(fb)
    r["mel.window"] = torch.hann_window(n_fft)
    r["enc.pe"] = sinusoids(cfg["enc_max_pos"], cfg["enc_dim"]).float()
    d = cfg["enc_dim"]

    def tlayers(src, dst, n_layers):
        for i in range(n_layers):
            s, o = f"{src}.layers.{i}.", f"{dst}.layers.{i}."
            r[o + "qkv.w"] = torch.cat([t(s + "self_attn.q_proj.weight"), t(s + "self_attn.k_proj.weight"),
                                        t(s + "self_attn.v_proj.weight")], 0).contiguous()
            r[o + "qkv.b"] = torch.cat([t(s + "self_attn.q_proj.bias"), torch.zeros(d), t(s + "self_attn.v_proj.bias")])
            r[o + "o.w"], r[o + "o.b"] = t(s + "self_attn.out_proj.weight"), t(s + "self_attn.out_proj.bias")
            r[o + "ln1.w"], r[o + "ln1.b"] = t(s + "self_attn_layer_norm.weight"), t(s + "self_attn_layer_norm.bias")
            r[o + "ln2.w"], r[o + "ln2.b"] = t(s + "final_layer_norm.weight"), t(s + "final_layer_norm.bias")
            r[o + "fc1.w"], r[o + "fc1.b"] = t(s + "fc1.weight"), t(s + "fc1.bias")
            r[o + "fc2.w"], r[o + "fc2.b"] = t(s + "fc2.weight"), t(s + "fc2.bias")

    for src, dst in (("semantic_encoder", "sem"), ("acoustic_encoder", "aco")):
        for cv in ("conv1", "conv2"):
            w = t(f"{src}.{cv}.weight")                                   # [out, in, 3] -> im2col [out][j*in+c]
            r[f"{dst}.{cv}.w"] = w.permute(0, 2, 1).reshape(w.shape[0], -1).contiguous()
            r[f"{dst}.{cv}.b"] = t(f"{src}.{cv}.bias")
        tlayers(src, dst, cfg["enc_layers"])
        r[f"{dst}.ln.w"], r[f"{dst}.ln.b"] = t(f"{src}.layer_norm.weight"), t(f"{src}.layer_norm.bias")
    tlayers("semantic_encoder_adapter", "semad", cfg["sem_adapter_layers"])
    r["semad.ln.w"], r["semad.ln.b"] = t("semantic_encoder_adapter.layer_norm.weight"), t("semantic_encoder_adapter.layer_norm.bias")
    r["prervq.proj.w"], r["prervq.proj.b"] = t("pre_rvq_adapter.proj.weight"), t("pre_rvq_adapter.proj.bias")
    tlayers("pre_rvq_adapter", "prervq", cfg["pre_rvq_layers"])
    r["prervq.ln.w"], r["prervq.ln.b"] = t("pre_rvq_adapter.layer_norm.weight"), t("pre_rvq_adapter.layer_norm.bias")
    for nm, role in (("gate_proj", "gate"), ("up_proj", "up")):
        w = t(f"downsample.{nm}.weight")                                  # [out, in, P] -> rows of P frames [out][j*in+c]
        r[f"down.{role}.w"] = w.permute(0, 2, 1).reshape(w.shape[0], -1).contiguous()
    r["down.down.w"] = t("downsample.down_proj.weight")
    r["down.ln.w"], r["down.ln.b"] = t("downsample.layer_norm.weight"), t("downsample.layer_norm.bias")
    g, v = t("quantizer.input_proj.weight_g"), t("quantizer.input_proj.weight_v")
    nrm = v.double().pow(2).sum(dim=(1, 2), keepdim=True).sqrt().float()
    r["rvq.in.w"] = (g * v / nrm)[:, :, 0].contiguous()
    r["rvq.in.b"] = t("quantizer.input_proj.bias")
    for q in range(cfg["nq"]):
        cb = t(f"quantizer.quantizers.{q}.codebook")
        r[f"rvq.cc.{q}"] = cb.pow(2).sum(1)                               # codebook.pow(2).sum(1) (quantizer.py:169)
    return r


def role_tensors(cfg, sd):
    """reference state dict (names of XY_Tokenizer.state_dict()) -> {role: fp32 tensor}.
    Every transformation is a pure re-layout except the weight-norm fold of quantizer.output_proj
    (w = g * v / ||v||, torch weight_norm dim=0) and the q/k/v concatenation."""
    t = lambda k: (torch.from_numpy(sd[k]) if isinstance(sd[k], np.ndarray) else sd[k]).detach().float().cpu()
    r = {}
    g, v = t("quantizer.output_proj.weight_g"), t("quantizer.output_proj.weight_v")
    nrm = v.double().pow(2).sum(dim=(1, 2), keepdim=True).sqrt().float()
    r["rvq.out.w"] = (g * v / nrm)[:, :, 0].contiguous()
    r["rvq.out.b"] = t("quantizer.output_proj.bias")
    for q in range(cfg["nq"]):
        r[f"rvq.codebook.{q}"] = t(f"quantizer.quantizers.{q}.codebook")

    def tlayers(src, dst, n_layers, d):
        for n in range(n_layers):
            s, o = f"{src}.layers.{n}.", f"{dst}.layers.{n}."
            r[o + "qkv.w"] = torch.cat([t(s + "self_attn.q_proj.weight"), t(s + "self_attn.k_proj.weight"),
                                        t(s + "self_attn.v_proj.weight")], 0).contiguous()
            r[o + "qkv.b"] = torch.cat([t(s + "self_attn.q_proj.bias"), torch.zeros(d), t(s + "self_attn.v_proj.bias")])
            r[o + "o.w"], r[o + "o.b"] = t(s + "self_attn.out_proj.weight"), t(s + "self_attn.out_proj.bias")
            r[o + "ln1.w"], r[o + "ln1.b"] = t(s + "self_attn_layer_norm.weight"), t(s + "self_attn_layer_norm.bias")
            r[o + "ln2.w"], r[o + "ln2.b"] = t(s + "final_layer_norm.weight"), t(s + "final_layer_norm.bias")
            r[o + "fc1.w"], r[o + "fc1.b"] = t(s + "fc1.weight"), t(s + "fc1.bias")
            r[o + "fc2.w"], r[o + "fc2.b"] = t(s + "fc2.weight"), t(s + "fc2.bias")

    r["adapter.proj.w"], r["adapter.proj.b"] = t("post_rvq_adapter.proj.weight"), t("post_rvq_adapter.proj.bias")
    r["adapter.pe"] = (t("post_rvq_adapter.positional_embedding") if "post_rvq_adapter.positional_embedding" in sd
                       else sinusoids(cfg["adapter_max_pos"], cfg["adapter_dim"]).float())
    tlayers("post_rvq_adapter", "adapter", cfg["adapter_layers"], cfg["adapter_dim"])
    r["adapter.ln.w"], r["adapter.ln.b"] = t("post_rvq_adapter.layer_norm.weight"), t("post_rvq_adapter.layer_norm.bias")
    r["adapter.out.w"], r["adapter.out.b"] = t("post_rvq_adapter.out_proj.weight"), t("post_rvq_adapter.out_proj.bias")
    # ConvTranspose1d weight [Cin, Cout, k] -> GEMM rows (j, cout): W'[j*Cout+o][ci]
    up = t("upsample.up_conv.weight")
    r["up.w"] = up.permute(2, 1, 0).reshape(-1, up.shape[0]).contiguous()
    r["dec.pe"] = (t("acoustic_decoder.positional_embedding") if "acoustic_decoder.positional_embedding" in sd
                   else sinusoids(cfg["dec_max_pos"], cfg["dec_dim"]).float())
    tlayers("acoustic_decoder", "dec", cfg["dec_layers"], cfg["dec_dim"])
    r["dec.ln.w"], r["dec.ln.b"] = t("acoustic_decoder.layer_norm.weight"), t("acoustic_decoder.layer_norm.bias")
    for nm in ("deconv1", "deconv2"):
        w = t(f"acoustic_decoder.{nm}.weight")
        r[f"dec.{nm}.w"] = w.permute(2, 1, 0).reshape(-1, w.shape[0]).contiguous()
        r[f"dec.{nm}.b"] = t(f"acoustic_decoder.{nm}.bias")
    # Conv1d weight [Cout, Cin, 7] -> im2col GEMM W'[o][j*Cin+c]
    we = t("enhanced_vocos.backbone.embed.weight")
    r["voc.embed.w"] = we.permute(0, 2, 1).reshape(we.shape[0], -1).contiguous()
    r["voc.embed.b"] = t("enhanced_vocos.backbone.embed.bias")
    r["voc.norm.w"], r["voc.norm.b"] = t("enhanced_vocos.backbone.norm.weight"), t("enhanced_vocos.backbone.norm.bias")
    for n in range(cfg["voc_layers"]):
        s, o = f"enhanced_vocos.backbone.convnext.{n}.", f"voc.blocks.{n}."
        r[o + "dw.w"] = t(s + "dwconv.weight")[:, 0, :].t().contiguous()          # [7][C]
        r[o + "dw.b"] = t(s + "dwconv.bias")
        r[o + "ln.w"], r[o + "ln.b"] = t(s + "norm.weight"), t(s + "norm.bias")
        r[o + "pw1.w"], r[o + "pw1.b"] = t(s + "pwconv1.weight"), t(s + "pwconv1.bias")
        r[o + "pw2.w"], r[o + "pw2.b"] = t(s + "pwconv2.weight"), t(s + "pwconv2.bias")
        r[o + "gamma"] = t(s + "gamma")
    r["voc.final_ln.w"] = t("enhanced_vocos.backbone.final_layer_norm.weight")
    r["voc.final_ln.b"] = t("enhanced_vocos.backbone.final_layer_norm.bias")
    r["voc.head.w"], r["voc.head.b"] = t("enhanced_vocos.head.out.weight"), t("enhanced_vocos.head.out.bias")
    r["istft.basis"] = istft_basis(cfg["n_fft"])
    r["istft.window"] = (t("enhanced_vocos.head.istft.window") if "enhanced_vocos.head.istft.window" in sd
                         else torch.hann_window(cfg["n_fft"]))
    return r


class CodecEngine:
    def __init__(self, cfg: dict, device="cuda:0"):
        if not torch.cuda.is_available():
            raise capi.MttsError("no GPU visible: the mtts codec only runs on MI355X (no CPU fallback)")
        from . import synth_codec
        full = synth_codec.codec_config()
        full.update(cfg)                       # configs written before the encode side existed lack its keys
        self.cfg = cfg = full
        self.device = torch.device(device)
        self.lib = capi.lib()
        c = MttsCodecConfig()
        for name, _ in MttsCodecConfig._fields_:
            setattr(c, name, int(cfg[name]))
        self._h = C.c_void_p()
        _check(self.lib.mtts_codec_create(C.byref(c), self.device.index or 0, C.byref(self._h)))

    def close(self):
        if self._h:
            self.lib.mtts_codec_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bind_state_dict(self, sd, encoder=None):
        roles = role_tensors(self.cfg, sd)
        if encoder is None:
            encoder = "semantic_encoder.conv1.weight" in sd
        if encoder:
            roles.update(encoder_role_tensors(self.cfg, sd))
        self.has_encoder = bool(encoder)
        for role, t in roles.items():
            d = t.to(device=self.device, dtype=torch.float32).contiguous()
            _check(self.lib.mtts_codec_bind(self._h, role.encode(), d.data_ptr(), d.numel(), None))
        torch.cuda.synchronize(self.device)

    def detokenize(self, codes: torch.Tensor, lens):
        """codes int64 [nq,B,T] on device; lens list[int] -> wav [B, T*1920] (device fp32)."""
        nq, B, T = codes.shape
        codes = codes.to(device=self.device, dtype=torch.int64).contiguous()
        hop_total = self.cfg["decoder_upsample_rate"]
        wav = torch.empty(B, T * hop_total, dtype=torch.float32, device=self.device)
        lens_arr = (C.c_int32 * B)(*[int(x) for x in lens])
        torch.cuda.synchronize(self.device)
        _check(self.lib.mtts_codec_detokenize(self._h, codes.data_ptr(), lens_arr, B, T, wav.data_ptr(), None))
        return wav

    def detokenize_async(self, codes: torch.Tensor, lens, stream):
        """Enqueue-only detokenize on `stream` (torch.cuda.Stream); call check(stream) before reading the result."""
        nq, B, T = codes.shape
        assert codes.is_contiguous() and codes.dtype == torch.int64
        wav = torch.empty(B, T * self.cfg["decoder_upsample_rate"], dtype=torch.float32, device=self.device)
        lens_arr = (C.c_int32 * B)(*[int(x) for x in lens])
        _check(self.lib.mtts_codec_detokenize_async(self._h, codes.data_ptr(), lens_arr, B, T, wav.data_ptr(),
                                                    C.c_void_p(stream.cuda_stream)))
        return wav

    def check(self, stream):
        _check(self.lib.mtts_codec_check(self._h, C.c_void_p(stream.cuda_stream)))

    def tokenize(self, wav: torch.Tensor, lens):
        """wav fp32 [B, n<=480000] on device (zero padded), lens -> (codes int64 [nq,B,375], code_lens)."""
        if not getattr(self, "has_encoder", False):
            raise capi.MttsError("encoder weights are not bound")
        B, n = wav.shape
        wav = wav.to(device=self.device, dtype=torch.float32).contiguous()
        Tc = self.cfg["mel_frames"] // (2 * self.cfg["down_pool"])
        codes = torch.zeros(self.cfg["nq"], B, Tc, dtype=torch.int64, device=self.device)
        lens_arr = (C.c_int32 * B)(*[int(x) for x in lens])
        out_lens = (C.c_int32 * B)()
        torch.cuda.synchronize(self.device)
        _check(self.lib.mtts_codec_tokenize(self._h, wav.data_ptr(), lens_arr, B, n, codes.data_ptr(), out_lens, None))
        return codes, [int(x) for x in out_lens]

    def encode(self, wav_list, overlap_seconds=10):
        """XY_Tokenizer.encode (reference model.py:131-192): 30 s windows, stride 30-overlap seconds."""
        c = self.cfg
        sr, ds = c["input_sample_rate"], c["encoder_downsample_rate"]
        chunk, dur = int(30 * sr), int((30 - overlap_seconds) * sr)
        code_dur = dur // ds
        B = len(wav_list)
        lens = [int(len(x)) for x in wav_list]
        maxlen = max(lens)
        wav = torch.zeros(B, maxlen, dtype=torch.float32, device=self.device)
        for i, x in enumerate(wav_list):
            wav[i, :lens[i]] = torch.as_tensor(x, dtype=torch.float32).to(self.device)
        lens_t = np.array(lens)
        outs = []
        for ch in range((maxlen + dur - 1) // dur):
            start = ch * dur
            end = min(start + chunk, maxlen)
            cl = np.clip(lens_t - start, 0, end - start)
            if cl.max() == 0:
                continue
            codes, clen = self.tokenize(wav[:, start:end], cl.tolist())
            blk = torch.zeros(c["nq"], B, code_dur, dtype=torch.int64, device=self.device)
            for b in range(B):
                v = min(clen[b], code_dur)
                if v > 0:
                    blk[:, b, :v] = codes[:, b, :v]
            outs.append(blk)
        if not outs:
            return [torch.zeros(c["nq"], 0, dtype=torch.long, device=self.device) for _ in range(B)]
        full = torch.cat(outs, dim=-1)
        return [full[:, i, :lens[i] // ds] for i in range(B)]

    def decode(self, codes_list, overlap_seconds=10):
        """XY_Tokenizer.decode (reference model.py:195-256): 30 s windows, keep 30-overlap seconds."""
        c = self.cfg
        duration = 30 - overlap_seconds
        chunk_len = int(30 * c["input_sample_rate"] // c["encoder_downsample_rate"])
        dur_len = int(duration * c["input_sample_rate"] // c["encoder_downsample_rate"])
        up = c["decoder_upsample_rate"]
        dur_wav = dur_len * up
        B = len(codes_list)
        maxT = max(int(x.shape[-1]) for x in codes_list)
        codes = torch.zeros(c["nq"], B, maxT, dtype=torch.int64, device=self.device)
        lens = []
        for i, x in enumerate(codes_list):
            x = torch.as_tensor(x)
            codes[:, i, :x.shape[-1]] = x.to(self.device)
            lens.append(int(x.shape[-1]))
        lens_t = np.array(lens)
        wavs = []
        for ch in range((maxT + dur_len - 1) // dur_len):
            start = ch * dur_len
            end = min(start + chunk_len, maxT)
            cl = np.clip(lens_t - start, 0, end - start)
            if cl.max() == 0:
                continue
            y = self.detokenize(codes[:, :, start:end], cl.tolist())
            out = torch.zeros(B, dur_wav, dtype=torch.float32, device=self.device)
            for b in range(B):
                k = int(min(cl[b] * up, dur_wav))
                if k > 0:
                    out[b, :k] = y[b, :k]
            wavs.append(out)
        if not wavs:
            return [torch.zeros(0, device=self.device) for _ in range(B)]
        full = torch.cat(wavs, dim=-1)
        return [full[i, :lens[i] * up] for i in range(B)]

    def decode_each(self, codes_list, overlap_seconds=10, windows_per_call=32):
        """Every sequence decoded as if it were alone in its call -- what the reference's process_batch does, one
        `spt.decode([codes])` per sample (generation_utils.py:434-450) -- but executed together: the 30 s windows of
        ALL sequences are flattened and run `windows_per_call` at a time.  `decode()` above keeps XY_Tokenizer.decode's
        batch semantics, where short sequences are zero-padded to the longest and the padded frames leak into a
        sequence's last frames through the convolutions (so a sample's waveform depends on what it is batched with);
        here only windows of EQUAL length share a call, so nothing is padded: all full windows (375 codes) go
        together, each ragged last window with the others of its length."""
        c = self.cfg
        duration = 30 - overlap_seconds
        chunk_len = int(30 * c["input_sample_rate"] // c["encoder_downsample_rate"])
        dur_len = int(duration * c["input_sample_rate"] // c["encoder_downsample_rate"])
        up = c["decoder_upsample_rate"]
        dur_wav = dur_len * up
        B = len(codes_list)
        codes = [torch.as_tensor(x).to(device=self.device, dtype=torch.int64) for x in codes_list]
        lens = [int(x.shape[-1]) for x in codes]
        by_len = {}
        for b in range(B):       # window w of a sequence of n codes covers codes [w*dur_len, min(w*dur_len + chunk_len, n))
            for w in range((lens[b] + dur_len - 1) // dur_len):
                by_len.setdefault(min(lens[b] - w * dur_len, chunk_len), []).append((b, w))
        outs = [torch.zeros(lens[b] * up, dtype=torch.float32, device=self.device) for b in range(B)]
        for cl, jobs in sorted(by_len.items(), reverse=True):
            for g in range(0, len(jobs), int(windows_per_call)):
                grp = jobs[g:g + int(windows_per_call)]
                blk = torch.stack([codes[b][:, w * dur_len:w * dur_len + cl] for b, w in grp], dim=1).contiguous()
                y = self.detokenize(blk, [cl] * len(grp))
                k = int(min(cl * up, dur_wav))
                for j, (b, w) in enumerate(grp):
                    outs[b][w * dur_wav:w * dur_wav + k] = y[j, :k]
        return outs
