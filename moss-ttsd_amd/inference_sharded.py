"""The reference's single-process inference flow (`inference.py:52-112`: load_model -> .to(device) -> process_batch ->
save) sharded over the GPUs of one node -- BASELINE configs[3]: "batch 256 synthetic dialogues sharded 8 ways over xGMI
(RCCL broadcast weights, gather audio)".

One process per GPU (`torchrun --nproc-per-node N inference_sharded.py --jsonl ...`, or plain `python` for one GPU):
  * `load_model_sharded`: rank 0 reads the checkpoints from disk (one reader instead of N), its state dicts travel to
    the other ranks as a few flat buckets over RCCL (`mtts.dist.broadcast_state_dict`), and every rank builds its own
    engines from what it received;
  * `process_batch_sharded`: the batch is dealt by estimated work (prompt tokens: text tokens + 12.5 codes per second of
    prompt audio), every rank runs the unchanged `generation_utils.process_batch` on its share -- dialogues are
    independent, no per-step collective -- and rank 0 receives the text records and the audio (`gather_audio`: one
    exact-size buffer per rank over its own xGMI link).  Rank 0 returns exactly what `process_batch` returns for the
    whole batch; the other ranks return (None, None).
With one process (no process group) both functions reduce to `load_model` / `process_batch`."""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import generation_utils as gu  # noqa: E402
from mtts import dist as mdist  # noqa: E402

MODEL_PATH = "fnlp/MOSS-TTSD-v0.5"
SYSTEM_PROMPT = ("You are a speech synthesizer that generates natural, realistic, and human-like conversational audio "
                 "from dialogue text.")
SPT_CONFIG_PATH = "XY_Tokenizer/config/xy_tokenizer_config.yaml"
SPT_CHECKPOINT_PATH = "XY_Tokenizer/weights/xy_tokenizer.ckpt"


def _world_rank():
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def init_distributed(backend=None):
    """Join the job `torchrun` started (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* in the environment): one process per
    GPU, backend nccl (= RCCL) on GPUs.  Returns (world, rank, device).  Without a launcher: (1, 0, cuda:0 | cpu)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    cuda = torch.cuda.is_available()
    if cuda and backend not in (None, "nccl"):
        local %= max(torch.cuda.device_count(), 1)       # gloo rehearsals: several ranks may share one card
    device = torch.device(f"cuda:{local}") if cuda else torch.device("cpu")
    if cuda:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or ("nccl" if cuda else "gloo")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    return (*_world_rank(), device)


def load_model_sharded(model_path, spt_config_path, spt_checkpoint_path, torch_dtype=torch.bfloat16,
                       attn_implementation="flash_attention_2", device=None, loader=None):
    """`generation_utils.load_model` for an N-rank job: -> (tokenizer, model, spt) on every rank, both models already
    on `device`.  `loader` (tests): callable returning (tokenizer, model, spt) in place of `gu.load_model`."""
    world, rank = _world_rank()
    loader = loader or gu.load_model
    if world == 1:
        tok, model, spt = loader(model_path, spt_config_path, spt_checkpoint_path, torch_dtype=torch_dtype,
                                 attn_implementation=attn_implementation)
        return tok, (model.to(device) if device is not None else model), (spt.to(device) if device is not None else spt)
    device = torch.device(device if device is not None else "cpu")
    tok = model = spt = None
    head = [None]
    if rank == 0:
        tok, model, spt = loader(model_path, spt_config_path, spt_checkpoint_path, torch_dtype=torch_dtype,
                                 attn_implementation=attn_implementation)
        gc = model.generation_config
        head = [{"cfg": model.config.to_dict(), "gen": dict(gc.__dict__), "dtype": model.dtype,
                 "spt_cfg": spt.cfg, "spt_rates": (spt.input_sample_rate, spt.output_sample_rate, spt.nq),
                 "tokenizer": tok}]
    dist.broadcast_object_list(head, src=0)
    h = head[0]
    ar_sd = dict(mdist.broadcast_state_dict(model._sd if rank == 0 else {}, device))
    codec_sd = dict(mdist.broadcast_state_dict(spt._sd if rank == 0 else {}, device))
    if rank != 0:
        from modeling_asteroid import AsteroidTTSInstruct, GenerationConfig
        from XY_Tokenizer.xy_tokenizer.model import XY_Tokenizer
        tok = h["tokenizer"]
        model = AsteroidTTSInstruct.from_state_dict(h["cfg"], ar_sd, GenerationConfig(**h["gen"]), dtype=h["dtype"])
        spt = XY_Tokenizer.from_engine_config(h["spt_cfg"], codec_sd, *h["spt_rates"])
    else:
        model._sd, spt._sd = ar_sd, codec_sd          # rank 0 binds from the device copies too
    return tok, model.eval().to(device), spt.eval().to(device)


def estimate_work(batch_items, tokenizer, system_prompt, use_normalize=False):
    """Per item: prompt tokens = tokens of the text prompt + 12.5 codes per second of prompt audio.  The number of frames
    a dialogue generates grows with its text, and its KV cost with prompt + generated tokens, so this is the deal's
    weight.  Reads wav HEADERS only (sample count / rate), no decoding or resampling."""
    out = []
    for item in batch_items:
        it = gu.process_jsonl_item(item)
        text = it["prompt_text"] + it["text"] if it["prompt_text"] else it["text"]
        if use_normalize:
            text = gu.normalize_text(text)
        n = len(tokenizer.encode(text.replace("[S1]", "<speaker1>").replace("[S2]", "<speaker2>")))
        audio = it["prompt_audio"]
        paths = list(audio.values()) if isinstance(audio, dict) else ([audio] if audio else [])
        for p in paths:
            try:
                if isinstance(p, tuple):
                    n += int(12.5 * p[0].shape[-1] / p[1])
                elif isinstance(p, str) and p:
                    n += int(12.5 * _wav_seconds(p))
            except Exception:
                pass                                    # unreadable audio fails inside process_batch, per sample
        out.append(n)
    return out


def _wav_seconds(path):
    import struct
    with open(path, "rb") as f:
        d = f.read(4096)
    i = 12
    rate = frame = None
    while i + 8 <= len(d):
        cid, sz = d[i:i + 4], struct.unpack("<I", d[i + 4:i + 8])[0]
        if cid == b"fmt ":
            _, ch, rate, _, frame, _ = struct.unpack("<HHIIHH", d[i + 8:i + 24])
        elif cid == b"data":
            return min(sz, os.path.getsize(path) - i - 8) / max(frame or 1, 1) / max(rate or 1, 1)
        i += 8 + sz + (sz & 1)
    return 0.0


def process_batch_sharded(batch_items, tokenizer, model, spt, device, system_prompt, start_idx, use_normalize=False,
                          run_local=None):
    """`generation_utils.process_batch` (reference generation_utils.py:341-473) over all ranks of the job.  Rank 0 returns
    (actual_texts_data, audio_results) for the WHOLE batch, in item order, in the reference's format; other ranks
    return (None, None).  A sample that fails on its rank is None in the result (the reference's convention); a
    rank-level failure raises on that rank.  `run_local` (tests): stands in for gu.process_batch."""
    world, rank = _world_rank()
    run_local = run_local or gu.process_batch
    if world == 1:
        return run_local(batch_items, tokenizer, model, spt, device, system_prompt, start_idx, use_normalize)
    work = estimate_work(batch_items, tokenizer, system_prompt, use_normalize)
    mine = mdist.shard_indices(work, world)[rank]
    print(f"[rank {rank}] {len(mine)} of {len(batch_items)} samples, estimated prompt tokens {sum(work[i] for i in mine)}")
    texts, audio = [], []
    if mine:
        # every row draws from the Philox stream of its position in the WHOLE batch: same tokens as the unsharded run
        prev, model.sample_rows = getattr(model, "sample_rows", None), list(mine)
        try:
            texts, audio = run_local([batch_items[i] for i in mine], tokenizer, model, spt, device, system_prompt,
                                     start_idx, use_normalize, indices=[start_idx + i for i in mine])
        finally:
            model.sample_rows = prev
    dev = torch.device(device)
    local = [(i, None if a is None else a["audio_data"].reshape(-1)) for i, a in zip(mine, audio)]
    got = mdist.gather_audio(local, dev if dist.get_backend() == "nccl" else torch.device("cpu"))
    all_texts = mdist.gather_objects(list(zip(mine, texts)))
    if rank != 0:
        return None, None
    n = len(batch_items)
    texts_out, audio_out = [None] * n, [None] * n
    for part in all_texts:
        for i, t in part:
            texts_out[i] = t
    for i, wav in got:
        if wav is not None:
            audio_out[i] = {"audio_data": wav.cpu().unsqueeze(0), "sample_rate": spt.output_sample_rate, "index": start_idx + i}
    return texts_out, audio_out


def main(argv=None):
    """`inference.py`'s command line (reference inference.py:17-112), one process per GPU."""
    ap = argparse.ArgumentParser(description="TTS inference with Asteroid model, batch-sharded over the node's GPUs")
    ap.add_argument("--jsonl", default="examples/examples.jsonl")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--output_dir", default="outputs")
    ap.add_argument("--summary_file", default=None)
    ap.add_argument("--use_normalize", action="store_true", default=False)
    ap.add_argument("--dtype", choices=["bf16", "fp16", "fp32"], default="bf16")
    ap.add_argument("--attn_implementation", choices=["flash_attention_2", "sdpa", "eager"], default="flash_attention_2")
    ap.add_argument("--model_path", default=MODEL_PATH)
    ap.add_argument("--spt_config", default=SPT_CONFIG_PATH)
    ap.add_argument("--spt_checkpoint", default=SPT_CHECKPOINT_PATH)
    args = ap.parse_args(argv)
    torch_dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    world, rank, device = init_distributed()
    if rank == 0:
        os.makedirs(args.output_dir, exist_ok=True)
        print(f"Using {world} rank(s), device: {device}, dtype: {args.dtype}")
    tokenizer, model, spt = load_model_sharded(args.model_path, args.spt_config, args.spt_checkpoint, torch_dtype=torch_dtype,
                                               attn_implementation=args.attn_implementation, device=device)
    with open(args.jsonl) as f:
        items = [json.loads(line) for line in f if line.strip()]
    if args.seed is not None:
        torch.manual_seed(args.seed)                     # every rank the same key; rows differ by their job-wide position
    texts, results = process_batch_sharded(items, tokenizer, model, spt, device, SYSTEM_PROMPT, 0, args.use_normalize)
    if rank == 0:
        if args.summary_file:
            with open(args.summary_file, "w", encoding="utf-8") as f:
                for t in texts:
                    f.write(json.dumps({"text": t["original_text"], "normalized_text": t["normalized_text"],
                                        "final_text": t["final_text"]}, ensure_ascii=False) + "\n")
        saved = 0
        for idx, res in enumerate(results):
            if res is None:
                print(f"Skipping sample {idx} due to generation error")
                continue
            path = os.path.join(args.output_dir, f"output_{idx}.wav")
            gu.save_wav(path, res["audio_data"], res["sample_rate"])
            print(f"Saved audio to {path}")
            saved += 1
        print(f"Inference completed. Saved {saved}/{len(items)} audio files to {args.output_dir}")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
