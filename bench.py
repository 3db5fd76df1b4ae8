#!/usr/bin/env python3
"""bench.py -- audio codec tokens/s of the MI355X-native AR decode path.

Workload (BASELINE.json configs[2], the one `metric` is quoted on): batch 32
synthetic dialogues per GPU, ASSUMED Qwen3-1.7B-class AsteroidTTS dims (SURVEY.md
reading notes), bf16, top-k/top-p sampling on all 8 channels, KV context ramped
to 4096 tokens per sequence; the timed region is K decode steps at that context
(the hardest point of the ramp), inputs (weights, KV pages) resident in HBM.
A "step" = one decode step of the whole batch = B frames = 8*B codec ids.

N>1: one process per GPU (torch.distributed, backend nccl = RCCL), dialogues are
independent, so the batch is sharded (32 per rank, weak scaling); the only
collectives are the weight broadcast at start-up (flat buckets), the metric reduction,
and -- outside the timed region, reported as `end_to_end` -- the gather of every rank's
decoded audio on rank 0 (BASELINE configs[3]).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "moss-ttsd_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes_per_step(cfg, B, L):
    """W + B*L*kv_bytes + B*(kv_bytes + 8*H*2)."""
    H, I, nl = cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"]
    nq, nkv, D = cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"]
    per_layer = H * (nq + 2 * nkv) * D + nq * D * H + 3 * H * I + 2 * H + 2 * D
    W = 2 * (nl * per_layer + H + cfg["vocab_size"] * H + 7 * cfg["speech_vocab_size"] * H)
    kv = 2 * nl * nkv * D * 2
    return W + B * L * kv + B * (kv + 8 * H * 2), W, kv


def make_weights_on_device(cfg, seed, device, rank, world):
    """Random-init weights of the architecture, generated on rank 0's GPU; at N>1 they reach the other ranks the way a
    checkpoint does in inference_sharded.load_model_sharded: mtts.dist.broadcast_state_dict, a few flat <= 1 GiB
    buckets over RCCL.  Yields (name, tensor)."""
    import torch
    from mtts import dist as mdist
    from mtts import synth
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    lo, hi = cfg["speech_token_range"]

    def gen():
        for name, shape, kind in synth.weight_shapes(cfg):
            t = torch.empty(shape, dtype=torch.bfloat16, device=device)
            if kind == "norm":
                t.copy_(1.0 + 0.1 * torch.randn(shape, device=device, generator=g))
            else:
                t.copy_(0.02 * torch.randn(shape, device=device, generator=g))
                if name.endswith("embedding_list.0.weight"):
                    # keep channel 0 inside the speech range so no dialogue flushes early
                    t[lo:hi] *= 8.0
            yield name, t

    if world == 1:
        yield from gen()                         # one tensor at a time: nothing to hold on to
    else:
        yield from mdist.broadcast_state_dict(dict(gen()) if rank == 0 else {}, device)


def cpu_baseline(cfg, B, L, layers_sampled=1, steps=2, seed=3):
    """The oracle (numpy restatement of the reference's eager CPU path) timed on this
    box's host cores on a bounded sample: `layers_sampled` of the decoder layers at
    full width + all 8 heads, batch B at KV length L, `steps` decode steps; the
    per-layer time is scaled to the full depth."""
    from oracle import asteroid_oracle as ao
    from mtts import synth
    small = dict(cfg)
    small["num_hidden_layers"] = layers_sampled
    rng = np.random.default_rng(seed)
    w = {}
    for name, shape, kind in synth.weight_shapes(small):
        if kind == "norm":
            w[name] = np.ones(shape, dtype=np.float32)
        else:
            a = rng.standard_normal(shape, dtype=np.float32)
            a *= np.float32(0.02)
            w[name] = ao.round_bf16(a)
    orc = ao.AsteroidOracle(small, w, "bf16")
    nkv, D = cfg["num_key_value_heads"], cfg["head_dim"]
    for n in range(layers_sampled):
        orc.K[n] = ao.round_bf16(rng.standard_normal((B, nkv, L - steps, D), dtype=np.float32))
        orc.V[n] = ao.round_bf16(rng.standard_normal((B, nkv, L - steps, D), dtype=np.float32))
    ids = np.stack([rng.integers(0, 1024, (B, 1)) for _ in range(8)], axis=-1)
    ids[..., 0] += 151665
    mask = np.ones((B, L), dtype=np.int64)
    t_layers, t_heads = [], []
    for s in range(steps):
        pos = np.full((B, 1), L - steps + s, dtype=np.int64)
        t0 = time.perf_counter()
        x = orc.forward_hidden(ids, pos, mask[:, :L - steps + s + 1])
        t1 = time.perf_counter()
        orc.heads(x)
        t2 = time.perf_counter()
        t_layers.append(t1 - t0)
        t_heads.append(t2 - t1)
    per_layer = min(t_layers) / layers_sampled
    step_time = per_layer * cfg["num_hidden_layers"] + min(t_heads)
    return step_time, per_layer, min(t_heads)


def host_cores():
    """CPU threads the numpy oracle can actually use here: BLAS pool size, capped by the affinity mask and the cgroup quota."""
    n = os.cpu_count() or 1
    try:
        import threadpoolctl
        info = threadpoolctl.threadpool_info()
        if info:
            n = max(i["num_threads"] for i in info)
    except Exception:
        pass
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def end_to_end_leg(eng, device, B, T_prompt, t_prefill, t_decode, steps_done, windows_per_call=32, keep=None):
    """End-to-end figure: prefill + every decode step run so far + codec decode of ALL the
    frames this run generated (full-depth XY_Tokenizer decoder, 30 s windows / 20 s stride as the reference)."""
    import torch
    from mtts import synth_codec
    from mtts.codec import CodecEngine
    gen = eng.read_generated(steps_done)                     # [G,B,8]
    G = gen.shape[0]
    n = G - 7
    codes = np.stack([gen[j:n + j, :, j] for j in range(8)], axis=0)      # un-shift (generation_utils.py:416-425)
    codes[0] -= 151665
    codes = np.clip(codes, 0, 1023).transpose(0, 2, 1)                    # [8,B,n]
    cfg = synth_codec.codec_config()
    cod = CodecEngine(cfg, device=str(device))
    cod.bind_state_dict(synth_codec.synth_weights(cfg, 5))
    t = torch.from_numpy(np.ascontiguousarray(codes)).to(device)
    cod.detokenize(t[:, :1, :375].contiguous(), [375])                    # warm-up / workspace
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total = 0
    for b0 in range(0, B, windows_per_call):
        wavs = cod.decode([t[:, b] for b in range(b0, min(B, b0 + windows_per_call))])
        total += sum(int(w.shape[0]) for w in wavs)
        if keep is not None:
            keep.extend(wavs)
    torch.cuda.synchronize()
    t_codec = time.perf_counter() - t0
    cod.close()
    audio_s = total / 24000.0
    wall = t_prefill + t_decode + t_codec
    return {"prefill_s": t_prefill, "decode_s": t_decode, "codec_s": t_codec, "frames": int(B * n),
            "audio_seconds": audio_s, "codec_ids_per_s": B * n * 8 / wall, "real_time_factor": audio_s / wall,
            "note": "decode_s = wall time of every decode step of this run (context ramp + timed steps, incl. host syncs)"}


def overlapped_leg(cfg, eng, device, ids, mask, max_length, layers, seed):
    """The same job end to end with the codec decoding finished 30 s windows on a second HIP stream while the
    decode loop keeps running (mtts/streaming.py)."""
    import torch
    from mtts import streaming, synth_codec
    from mtts.codec import CodecEngine
    ccfg = synth_codec.codec_config()
    cod = CodecEngine(ccfg, device=str(device))
    cod.bind_state_dict(synth_codec.synth_weights(ccfg, 5))
    cod.detokenize(torch.zeros(8, ids.shape[0], 375, dtype=torch.int64, device=device), [375] * ids.shape[0])   # workspace
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gen, wavs = streaming.generate_with_overlapped_decode(eng, cod, ids, mask, max_length, layers=layers,
                                                          do_samples=[True] * 8, seed=seed)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    cod.close()
    audio_s = sum(int(w.shape[0]) for w in wavs) / 24000.0
    frames = int((gen.shape[0] - 7) * gen.shape[1])
    return {"wall_s": wall, "frames": frames, "audio_seconds": audio_s, "codec_ids_per_s": frames * 8 / wall,
            "real_time_factor": audio_s / wall}


def codec_leg(device, windows=8, T=375, reps=3):
    """Secondary figure (not `value`): full-depth XY_Tokenizer decoder, `windows` 30 s windows per call."""
    import torch
    from mtts import synth_codec
    from mtts.codec import CodecEngine
    cfg = synth_codec.codec_config()
    eng = CodecEngine(cfg, device=str(device))
    eng.bind_state_dict(synth_codec.synth_weights(cfg, 5))
    codes = torch.randint(0, 1024, (cfg["nq"], windows, T), device=device)
    eng.detokenize(codes, [T] * windows)          # warm-up (allocates the workspace)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.detokenize(codes, [T] * windows)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    eng.close()
    flop = 1.14e12 * windows * (T / 375.0)        # ~1.14 TFLOP per 375-code window
    # the decoder's GEMMs run as 3 bf16 MFMAs per fp32 product (csrc/codec.hip: gemm_b3_kernel): the matrix-core
    # ceiling for this leg is a third of the dense bf16 peak; the exact-f32 MFMA peak is quoted for scale
    return {"ms_per_window": dt / windows * 1e3, "audio_seconds_per_s": windows * T * 0.08 / dt,
            "tflops_algorithmic": flop / dt / 1e12, "frac_of_bf16x3_mfma_peak": 3.0 * flop / dt / 2500e12,
            "vs_f32_mfma_peak": flop / dt / 157.3e12, "windows_per_call": windows, "codes_per_window": T}


def launch_ranks(n):
    """Start `n` ranks of this script (torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1) as a child
    process tree and return its exit code.  The caller has not touched the GPU, and nothing is exec'd over it."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def dry_run(args, dist, world, rank):
    """Everything of an N-rank run except the engine: rendezvous, barrier, MAX/SUM reductions, one JSON line from
    rank 0.  Every rank pretends its K steps took K x 4 ms."""
    import torch
    B, K = args.batch, args.steps
    if world > 1:
        dist.barrier()
    tmax = torch.tensor([K * 4e-3 * (1.0 + 0.01 * rank)], dtype=torch.float64)
    units = torch.tensor([float(B * K * 8)], dtype=torch.float64)
    ranks = torch.zeros(world, dtype=torch.int64)
    ranks[rank] = 1
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(units, op=dist.ReduceOp.SUM)
        dist.all_reduce(ranks, op=dist.ReduceOp.SUM)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "audio codec tokens/sec/node (decode, bf16, batch 32 @ 4k ctx)",
                          "value": float(units.item()) / float(tmax.item()), "unit": "codec_tokens/s",
                          "n_gpus": world, "ranks": int(ranks.sum().item()), "backend": "gloo", "steps": K,
                          "warmup": args.warmup, "ms_per_step": float(tmax.item()) / K * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
                          "invalid": "dry run: launcher and collectives only, no engine"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def codec_cpu_baseline(codes_n=100):
    """The codec oracle (numpy restatement of XY_Tokenizer.decode, full depth) on the host cores, one sequence of
    `codes_n` codes (8 s of audio): the CPU figure beside the codec leg."""
    from mtts import synth_codec
    from oracle import codec_oracle as co
    cfg = synth_codec.codec_config()
    orc = co.CodecOracle(cfg, synth_codec.synth_weights(cfg, 5))
    codes = np.random.default_rng(0).integers(0, 1024, (cfg["nq"], codes_n))
    t0 = time.perf_counter()
    y = orc.decode([codes])[0]
    dt = time.perf_counter() - t0
    return {"value": (y.shape[0] / 24000.0) / dt, "unit": "audio_seconds/s", "cores": host_cores(), "kind": "port",
            "sample": f"numpy codec oracle, full depth, one sequence of {codes_n} codes ({codes_n * 0.08:.1f} s of audio) in {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=32, help="dialogues per GPU")
    ap.add_argument("--context", type=int, default=4096)
    ap.add_argument("--prompt", type=int, default=512)
    ap.add_argument("--layers", type=int, default=0, help="override depth (debug only; result is marked invalid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=8)
    ap.add_argument("--no-codec", action="store_true")
    ap.add_argument("--greedy", action="store_true",
                    help="argmax on every channel instead of top-k/top-p sampling (BASELINE configs[1]: --batch 1 --context 2048 --greedy)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / collective rehearsal without the engine (CPU, gloo): every rank reports a fixed "
                         "synthetic step time; checks that --gpus N really runs N ranks (the result is marked invalid)")
    ap.add_argument("--fake-context", action="store_true",
                    help="PMC/profiling runs only: jump to the target context with mtts_debug_set_kv_len instead of "
                         "ramping (cache content is not meaningful; the result is marked invalid)")
    args = ap.parse_args()

    # `python bench.py --gpus N` on its own (no launcher around it): start N fresh ranks, one per GPU, before this
    # process makes any GPU call, and hand its exit code back.  Under torchrun (WORLD_SIZE set) this IS one of the ranks.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): "
                         "refusing to report a number for a different job size")
    backend = "gloo" if args.dry_run else os.environ.get("MTTS_BENCH_BACKEND", "nccl")   # "gloo" on GPUs only to rehearse N>1 on a 1-GPU box
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dry_run:
            dist.init_process_group("gloo")
        else:
            if backend != "nccl":
                local = local % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local)
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
            else:
                dist.init_process_group(backend)
        assert dist.get_world_size() == args.gpus
    if args.dry_run:
        return dry_run(args, dist, world, rank)
    from mtts import synth
    from mtts.engine import Engine
    device = torch.device(f"cuda:{local}")
    torch.cuda.set_device(device)

    cfg = synth.assumed_1p7b()
    if args.layers:
        cfg["num_hidden_layers"] = args.layers
    B, K, W, L = args.batch, args.steps, args.warmup, args.context
    T = args.prompt
    n_real = T - 7
    ramp = L - n_real - (K + W) - args.profile_steps
    assert ramp >= 0, "context too small for prompt + steps"

    t_setup = time.perf_counter()
    eng = Engine(cfg, max_batch=B, max_seq_len=L + 64, device=str(device))
    for name, t in make_weights_on_device(cfg, 1234, device, rank, world):
        eng.bind(name, t)
        del t
    from mtts import capi
    capi.check(eng.lib.mtts_weights_ready(eng._h))
    ids, mask = synth.synth_prompts(cfg, 77 + rank, B, T, audio_frac=0.5, ragged=False)
    layers = [dict(top_k=50, top_p=0.95, temperature=1.0, repetition_penalty=1.0)] * 8
    max_length = T + (L - n_real) + 8
    H_, I_, nl_ = cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"]
    stack_params = nl_ * (H_ * (cfg["num_attention_heads"] + 2 * cfg["num_key_value_heads"]) * cfg["head_dim"]
                          + cfg["num_attention_heads"] * cfg["head_dim"] * H_ + 3 * H_ * I_)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    do_samples = [not args.greedy] * 8
    eng.begin(ids, mask, max_length, layers=None if args.greedy else layers, do_samples=do_samples, seed=42 + rank)
    eng.sync_state()
    t_prefill = time.perf_counter() - t0
    # ramp the KV context (untimed for the headline, reported as ramp_frames_per_s)
    t0 = time.perf_counter()
    done_steps = 0
    if args.fake_context:
        eng.debug_set_kv_len(n_real + ramp)
        done_steps = ramp
    while done_steps < ramp:
        n = min(256, ramp - done_steps)
        eng.step(n)
        done_steps += n
        st, fin = eng.sync_state()
        assert not fin, "a dialogue finished during the ramp: synthetic weights must keep channel 0 in the speech range"
    t_ramp = time.perf_counter() - t0
    eng.step(W)
    eng.sync_state()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.step(K)
    st, fin = eng.sync_state()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    assert not fin and st == (0 if args.fake_context else ramp) + W + K, (st, fin)
    # roofline leg: per-kernel HIP events on the launch stream for a few more steps at the same context
    eng.profile(True)
    eng.step(args.profile_steps)
    eng.sync_state()
    prof = {}
    for which, nm in ((0, "attn_scores_kernel"), (1, "attn_pv_kernel"), (3, "decode_step")):
        ms, n, by = eng.profile_read(which)
        prof[nm] = dict(ms=ms, launches=n, bytes=by)
    eng.profile(False)
    # dominant-kernel duration for the roofline: a train of back-to-back launches of each attention pass at the
    # state the timed region ended in (KV length ~L), two HIP events on the launch stream
    for ph, nm in ((1, "attn_scores_kernel"), (2, "attn_pv_kernel")):
        ms, by = eng.attn_bench(ph, 4 * cfg["num_hidden_layers"])
        prof[nm].update(train_ms=ms, train_bytes=by)
    t_decode_all = t_ramp + dt
    steps_all, _ = eng.sync_state()
    try:                       # sealed KV pages: how many of this run's complete pages sealed, which layers read them
        kv_pack = eng.kv_pack_stats()
    except Exception:          # engine without sealed pages (MTTS_KV_PACK=0)
        kv_pack = None

    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    units = torch.tensor([float(B * K * 8)], dtype=torch.float64, device=device)
    nranks = torch.ones(1, dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(units, op=dist.ReduceOp.SUM)
        dist.all_reduce(nranks, op=dist.ReduceOp.SUM)
    dt_max = float(tmax.item())
    total_ids = float(units.item())

    # N>1 (BASELINE configs[3]: "RCCL broadcast weights, gather audio"): every rank decodes the frames it generated and
    # rank 0 receives all the audio (mtts.dist.gather_audio: one exact-size buffer per rank over its own xGMI link)
    e2e_sharded = None
    e2e_hung = False
    if world > 1 and not args.no_codec and not args.fake_context:
        # The leg runs on a helper thread with a deadline: the headline line must be printed even if the codec or the
        # audio exchange (RCCL point-to-point, never run on hardware before the first 8-GPU bench) gets stuck.
        import threading
        box = {}

        def leg():
            try:
                torch.cuda.set_device(device)
                from mtts import dist as mdist
                wavs = []
                r = end_to_end_leg(eng, device, B, T, t_prefill, t_decode_all, steps_all, keep=wavs)
                dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                # (gloo rehearsals on a 1-GPU box exchange through host memory: gloo has no device send / recv)
                got = mdist.gather_audio([(rank * B + i, w) for i, w in enumerate(wavs)],
                                         device if backend == "nccl" else torch.device("cpu"))
                torch.cuda.synchronize()
                dist.barrier()
                t_gather = time.perf_counter() - t0
                tm = torch.tensor([r["prefill_s"], r["decode_s"], r["codec_s"], t_gather], dtype=torch.float64, device=device)
                sm = torch.tensor([r["audio_seconds"], float(r["frames"])], dtype=torch.float64, device=device)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                dist.all_reduce(sm, op=dist.ReduceOp.SUM)
                if rank == 0:
                    assert len(got) == world * B and all(w is not None for _, w in got)
                    gathered = sum(int(w.numel()) for _, w in got)
                    assert abs(gathered / 24000.0 - float(sm[0])) < 1e-3 * float(sm[0])
                    wall = float(tm.sum())
                    box["out"] = {"prefill_s": float(tm[0]), "decode_s": float(tm[1]), "codec_s": float(tm[2]), "gather_s": float(tm[3]),
                                  "gathered_bytes": gathered * 4, "gather_GBps_into_rank0": gathered * 4 / max(float(tm[3]), 1e-9) / 1e9,
                                  "frames": int(sm[1]), "audio_seconds": float(sm[0]), "codec_ids_per_s": float(sm[1]) * 8 / wall,
                                  "real_time_factor": float(sm[0]) / wall,
                                  "note": "slowest rank's prefill + decode (context ramp + timed steps) + codec decode of all its frames, "
                                          "then the audio of every rank gathered on rank 0"}
                box["done"] = True
            except Exception as ex:          # noqa: BLE001
                box["err"] = repr(ex)

        th = threading.Thread(target=leg, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("MTTS_BENCH_E2E_DEADLINE", "240")))
        if th.is_alive():
            e2e_hung = True
            e2e_sharded = {"error": "the sharded end-to-end leg (codec decode + audio gather) did not finish before its deadline; "
                                    "the headline figures above do not depend on it"}
        elif "err" in box:
            e2e_sharded = {"error": box["err"]}
        else:
            e2e_sharded = box.get("out")

    if rank == 0:
        Lt = L - args.profile_steps - K // 2          # mean KV length inside the timed region
        step_bytes, Wb, kvb = algorithmic_bytes_per_step(cfg, B, Lt)
        value = total_ids / dt_max
        frames = value / 8.0
        ms_step = dt_max / K * 1e3
        dom = max(("attn_scores_kernel", "attn_pv_kernel"), key=lambda k: prof[k]["train_ms"])
        p = prof[dom]
        avg_ms = p["train_ms"]
        bytes_per_launch = p["train_bytes"]
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM traffic per launch from separate --pmc passes of the kernels as they are NOW (tools/pmc_summary.py; FETCH_SIZE
        # doubled per the gfx950 correction).  Measured at B=32, L~4095: only quoted for that workload, and only from the
        # file of the round that last touched csrc/attn.hip (r03: sealed pages, template arguments <G, fused, sealed>).
        traffic, traffic_src = None, None
        sealed = bool(kv_pack and kv_pack["k_layers_on"] == cfg["num_hidden_layers"] and kv_pack["v_layers_on"] == cfg["num_hidden_layers"])
        for cand in ("r03_pmc_attention.json",):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", cand)))["kernels"]
                if B == 32 and L == 4096 and not args.layers and sealed:
                    # the file holds the variants that run at this size (GQA group 2; template arguments after the group:
                    # fused q/k/v epilogue, sealed pages, ...): the full-context launches are the ones with the most calls
                    keys = [k for k in pmc if k.startswith(dom + "<2,")]
                    key = max(keys, key=lambda k: pmc[k]["launches"])
                    traffic = pmc[key]["hbm_bytes_per_launch"]
                    traffic_src = "profiles/" + cand + " : " + key
                    break
            except Exception:
                traffic = None
        out = {
            "metric": "audio codec tokens/sec/node (decode, bf16, batch 32 @ 4k ctx)",
            "value": value, "unit": "codec_tokens/s", "n_gpus": world, "ranks": int(nranks.item()),
            "backend": (backend + (" (RCCL)" if backend == "nccl" else "")) if world > 1 else None, "steps": K, "warmup": W,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic (random-init weights of the ASSUMED 1.7B dims, synthetic prompts)",
            "config": {"workload": ("configs[2]: batch 32 synthetic dialogues/GPU, 4k-token KV context, top-k/top-p sampling on 8 channels, decode steps at full context"
                                    if (B == 32 and L == 4096 and not args.greedy) else
                                    f"batch {B} synthetic dialogues/GPU, {L}-token KV context, {'greedy' if args.greedy else 'top-k/top-p sampling'} on 8 channels, decode steps at full context"
                                    + (" (BASELINE configs[1])" if (B == 1 and L == 2048 and args.greedy) else " (not the headline configuration)")),
                       "batch_per_gpu": B, "context": L, "prompt": T, "layers": cfg["num_hidden_layers"],
                       "hidden": cfg["hidden_size"], "parallelism": f"dp{world} (batch shard, no per-step collective)"},
            "frames_per_s": frames, "real_time_factor": frames / 12.5,
            "step_algorithmic_bytes": step_bytes,
            "step_hbm_roofline_frac": (step_bytes / (dt_max / K)) / 1e9 / HBM_PEAK_GBS,
            "ramp_frames_per_s": B * ramp / t_ramp if ramp else None,
            "prefill_s": t_prefill, "setup_s": time.perf_counter() - t_setup,
            # prompt pass, compute-bound side: 2 * (decoder-stack parameters) * prompt tokens (the heads run on the last
            # token of each dialogue only), against the dense bf16 MFMA peak
            "prefill": {"tokens": int(B * n_real), "tokens_per_s": B * n_real / t_prefill,
                        "tflops": 2.0 * stack_params * B * n_real / t_prefill / 1e12, "mfma_peak_tflops": 2500.0},
            "roofline": {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "launches": 4 * cfg["num_hidden_layers"],
                         "how": "train of back-to-back launches at the end-of-run KV length, 2 HIP events on the launch stream; "
                                "`achieved` counts the ALGORITHMIC bytes (the bf16 K or V of every cached token, SURVEY 8d); the kernel "
                                "reads complete pages in their sealed 13-bit form (kv_pack), so `traffic` is below them"},
            "kv_pack": kv_pack,
            "kernels": {k: {"avg_ms": v["ms"] / max(v["launches"], 1), "launches": v["launches"], "train_ms": v.get("train_ms"),
                            "GBps": (v["bytes"] / max(v["launches"], 1)) / (v["ms"] / max(v["launches"], 1) * 1e-3) / 1e9
                            if v["ms"] > 0 and v["bytes"] else None} for k, v in prof.items()},
        }
        if args.layers:
            out["invalid"] = "depth overridden with --layers"
        if args.fake_context:
            out["invalid"] = "context faked with mtts_debug_set_kv_len (profiling run)"
        if e2e_sharded is not None:
            out["end_to_end"] = e2e_sharded
        if world == 1 and not args.no_codec:
            if not args.fake_context:
                out["end_to_end"] = end_to_end_leg(eng, device, B, T, t_prefill, t_decode_all, steps_all)
                if not args.greedy:
                    out["end_to_end_overlapped"] = overlapped_leg(cfg, eng, device, ids, mask, T + (L - n_real), layers, 42)
            eng.close()
            out["codec_decode"] = codec_leg(device)
            # process_batch hands the decoder up to 32 equal-length windows per call: the same leg at that size
            out["codec_decode"]["ms_per_window_at_32_per_call"] = codec_leg(device, windows=32, reps=2)["ms_per_window"]
            if not args.no_cpu_baseline:
                out["codec_decode"]["cpu_baseline"] = codec_cpu_baseline()
        if world == 1 and not args.no_cpu_baseline:
            cores = host_cores()
            st_time, per_layer, heads_t = cpu_baseline(cfg, B, L)
            out["cpu_baseline"] = {"value": B * 8 / st_time, "unit": "codec_tokens/s", "cores": cores, "kind": "port",
                                   "sample": f"numpy oracle (a port, slower than the reference's torch CPU path), EXTRAPOLATED: 1 of "
                                             f"{cfg['num_hidden_layers']} layers at full width + 8 heads, "
                                             f"batch {B} at KV length {L}, best of 2 decode steps; per-layer time "
                                             f"({per_layer:.3f} s) scaled to full depth, heads {heads_t:.3f} s"}
            # the REAL reference (AsteroidTTSInstruct.forward, eager and SDPA) timed in the build container by
            # tools/cpu_reference_timing.py -- it cannot travel to this box; quoted beside the port's figure
            try:
                rc = json.load(open(os.path.join(ROOT, "profiles", "r03_cpu_reference.json")))
                if B == rc["batch"] and L == rc["kv_len"]:
                    out["cpu_baseline"]["reference_container"] = {
                        "kind": "reference", "cores": rc["cores"], "torch": rc["torch"], "unit": "codec_tokens/s",
                        "eager": rc["eager"]["codec_ids_per_s"], "sdpa": rc["sdpa"]["codec_ids_per_s"],
                        "s_per_step_eager": rc["eager"]["s_per_step_28_layers"], "s_per_step_sdpa": rc["sdpa"]["s_per_step_28_layers"],
                        "sample": "reference forward, bf16, one decode step at B=32 / KV 4096 with 2 and 4 layers at the ASSUMED width "
                                  "-> fixed + 28 x per-layer; measured in the build container, not on this box",
                        "source": "profiles/r03_cpu_reference.json"}
                    if "codec_decode" in out:
                        out["codec_decode"].setdefault("cpu_baseline", {})["reference_container"] = {
                            "kind": "reference", "cores": rc["cores"], "value": rc["codec_decode"]["audio_s_per_s"],
                            "unit": "audio_seconds/s", "sample": "XY_Tokenizer.decode, full depth, one 375-code window"}
            except Exception:
                pass
        print(json.dumps(out), flush=True)
    if e2e_hung:
        os._exit(0)            # a stuck exchange must not keep the job (and the line already printed) hostage
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
